#!/bin/bash
# round 5, seventh collection: full suite, bench, exact fallback A/B with the prefetch, 2-rank gloo bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05g}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_suite.txt 2>&1; tail -5 $OUT/${TAG}_gpu_suite.txt
timeout 600 python tools/exact_group_bench.py > $OUT/${TAG}_exact_group_bench.txt 2>&1; cat $OUT/${TAG}_exact_group_bench.txt
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 400 $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
PRAG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_2ranks_gloo.json 2> $OUT/${TAG}_bench_2ranks_gloo.err; echo "gloo rc=$?"; tail -c 1500 $OUT/${TAG}_bench_2ranks_gloo.json
