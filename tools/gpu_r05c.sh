#!/bin/bash
# round 5, third collection: centred queries (per-row bias) - tests + embedding probe; tail overlap and sample A/B at the shard size
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05c}
timeout 900 python -m pytest tests/test_gpu_shadow.py tests/test_gpu_tiled8.py -x -q > $OUT/${TAG}_shadow_tests.txt 2>&1; tail -5 $OUT/${TAG}_shadow_tests.txt
timeout 600 python tools/embedding_probe.py 1048576 4194304 > $OUT/${TAG}_embedding_probe.txt 2> $OUT/${TAG}_embedding_probe.err; python - <<PY
import json
for l in open("$OUT/${TAG}_embedding_probe.txt"):
    r = json.loads(l)
    print(r["rows"], r["metric"], r["store"], r["queries"], r["structure"], "direct %.3f (fb %d) two-level %.3f ms fb %d surv %s" % (
        r["rows_scanned_directly"]["ms_per_search"], r["rows_scanned_directly"]["exact_fallbacks_last_search"], r["two_level"]["ms_per_search"], r["two_level"]["exact_fallbacks_last_search"],
        r["two_level"].get("survivors")))
PY
tail -3 $OUT/${TAG}_embedding_probe.err
for sw in "" "SHARD_TAIL=1" "PRAG_SHADOW_SAMPLE=2" "PRAG_SHADOW_SAMPLE=2 SHARD_TAIL=1" "PRAG_SHADOW_SAMPLE=1 SHARD_TAIL=1" "PRAG_SHADOW_AFFINE=0"; do
  echo "== $sw" >> $OUT/${TAG}_shard_ab.txt
  env $sw SHARD_REPS=300 timeout 200 python tools/shard_pass.py 2>&1 | grep "shard pass" | cut -c1-120 >> $OUT/${TAG}_shard_ab.txt
done
cat $OUT/${TAG}_shard_ab.txt
