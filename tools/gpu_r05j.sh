#!/bin/bash
# round 5, tenth collection: full suite; fused pass with the adaptive gather launch (gate fold off): shard A/B, bench, kernel stats
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05j}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_suite.txt 2>&1; tail -4 $OUT/${TAG}_gpu_suite.txt
rm -f $OUT/${TAG}_shard_ab.txt
for sw in "" "SHARD_FUSED=1" "PRAG_GATHER=1" "PRAG_GATHER=1 SHARD_FUSED=1" "SHARD_TAIL=1" "SHARD_FUSED=1"; do
  echo "== $sw" >> $OUT/${TAG}_shard_ab.txt
  env $sw SHARD_REPS=400 timeout 200 python tools/shard_pass.py 2>&1 | grep "shard pass" | cut -c1-60 >> $OUT/${TAG}_shard_ab.txt
done
cat $OUT/${TAG}_shard_ab.txt
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 300 $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
timeout 900 python bench.py --overlap-gate 3 --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_fused.json 2>> $OUT/${TAG}_bench.err; python - <<PY
import json
for n in ("bench", "bench_fused"):
    r = json.loads(open("$OUT/${TAG}_%s.json" % n).read().strip().splitlines()[-1])
    print(n, "ms_per_pass", r["config"]["ms_per_pass"], "gate_overlap", r["config"]["gate_overlap"][:40], "shard", r["config"].get("shard_pass_ms"), r["config"].get("shard_pass_mode"), r["config"].get("shard_pass_ms_by_mode"))
PY
cd /tmp && export TMPDIR=/tmp
SHARD_FUSED=1 SHARD_REPS=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_fused_stats -- python3 $R/tools/shard_pass.py > $OUT/${TAG}_fused_pass_under_rocprof.txt 2>&1
f=$(ls $OUT/${TAG}_fused_stats/*/*kernel_stats.csv | head -1); cp $f $OUT/${TAG}_fused_pass_kernel_stats.csv; head -10 $OUT/${TAG}_fused_pass_kernel_stats.csv | cut -c1-150
rm -rf $OUT/${TAG}_fused_stats
