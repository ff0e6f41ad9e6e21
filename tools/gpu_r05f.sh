#!/bin/bash
# round 5, sixth collection: full suite on retry tier + grouped exact scan; exact fallback A/B; fuzzers (short)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05f}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_suite.txt 2>&1; tail -5 $OUT/${TAG}_gpu_suite.txt
timeout 600 python tools/exact_group_bench.py > $OUT/${TAG}_exact_group_bench.txt 2>&1; cat $OUT/${TAG}_exact_group_bench.txt
timeout 150 python tools/fuzz_shadow.py 120 > $OUT/${TAG}_fuzz_shadow.txt 2>&1; tail -3 $OUT/${TAG}_fuzz_shadow.txt
timeout 100 python tools/fuzz_paths.py 80 > $OUT/${TAG}_fuzz_paths.txt 2>&1; tail -3 $OUT/${TAG}_fuzz_paths.txt
