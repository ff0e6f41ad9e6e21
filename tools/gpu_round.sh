#!/bin/bash
# GPU-box helper: the standard collection of a round under one tag.
#   tools/gpu_round.sh <tag> [suite|nosuite]
# -> gpurun_out/<tag>_gpu_suite.txt, <tag>_bench.json, <tag>_bench_shard_2625000.json, <tag>_bench_2ranks_gloo.json
set -u
TAG=$1; SUITE=${2:-suite}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [ "$SUITE" = suite ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_suite.txt 2>&1; tail -3 $OUT/${TAG}_gpu_suite.txt
fi
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 600 $OUT/${TAG}_bench.json
timeout 300 python bench.py --docs 2625000 --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_shard_2625000.json 2> $OUT/${TAG}_bench_shard.err
timeout 300 python bench.py --docs 2625000 --gate-batch 512 --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_shard_2625000_gate512.json 2>> $OUT/${TAG}_bench_shard.err
PRAG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_2ranks_gloo.json 2> $OUT/${TAG}_bench_2ranks_gloo.err; echo "gloo rc=$?"
python - <<PY
import json
for n in ("bench", "bench_shard_2625000", "bench_shard_2625000_gate512", "bench_2ranks_gloo"):
    try:
        r = json.loads(open("$OUT/${TAG}_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "ms_per_pass", r["config"]["ms_per_pass"], "scan", r["roofline"]["avg_launch_ms"], r["roofline"]["frac"],
              "gate", r["roofline_gate"]["avg_launch_ms"], "ranks", r.get("rccl_ranks"), r.get("backend"))
    except Exception as e:
        print(n, "unreadable:", e)
PY
