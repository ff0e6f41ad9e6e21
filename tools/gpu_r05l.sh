#!/bin/bash
# round 5: shadow chunks in MFMA operand order (no LDS staging in scan8) against the row-major chunks of rounds 2-4
# (libprag_ab.so = -DPRAG_SHADOW_CHUNK_MAJOR), same box, alternating
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05l}
timeout 900 python -m pytest tests/test_gpu_shadow.py tests/test_gpu_tiled8.py tests/test_gpu_index.py -m gpu -x -q > $OUT/${TAG}_tests.txt 2>&1; tail -4 $OUT/${TAG}_tests.txt
AB=$R/probing-rag_amd/lib/libprag_ab.so
rm -f $OUT/${TAG}_ab.txt
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export PRAG_LIB=$AB; else unset PRAG_LIB; fi
    timeout 600 python bench.py --no-cpu-baseline --measure-traffic 0 > $OUT/${TAG}_bench_${lib}_${rep}.json 2> $OUT/${TAG}_bench_${lib}_${rep}.err
    python - <<PY >> $OUT/${TAG}_ab.txt
import json
r = json.loads(open("$OUT/${TAG}_bench_${lib}_${rep}.json").read().strip().splitlines()[-1])
v = r.get("variants", {})
def ms(n):
    x = v.get(n) or {}
    return x.get("ms_per_search") or x.get("ms") or x.get("error")
print("$lib $rep", "pass", round(r["config"]["ms_per_pass"], 4), "scan8", round(r["roofline"]["avg_launch_ms"], 4), round(r["roofline"]["frac"], 4),
      "shard", r["config"].get("shard_pass_ms"), "| " + " ".join("%s=%s" % (k.replace("f16_cos_k10_", ""), ms(k)) for k in v if k.startswith("f16_cos_k10_q") or "1000" in k or "embedding" in k))
PY
  done
done
unset PRAG_LIB
cat $OUT/${TAG}_ab.txt
