#!/bin/bash
# round 5, first collection: suite + bench + embedding-shaped "before" numbers
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05a}
timeout 300 python tools/embedding_probe.py 1048576 > $OUT/${TAG}_embedding_probe.txt 2> $OUT/${TAG}_embedding_probe.err; tail -3 $OUT/${TAG}_embedding_probe.txt | cut -c1-600
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_suite.txt 2>&1; tail -5 $OUT/${TAG}_gpu_suite.txt
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 1500 $OUT/${TAG}_bench.json; tail -5 $OUT/${TAG}_bench.err
