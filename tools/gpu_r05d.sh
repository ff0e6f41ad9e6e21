#!/bin/bash
# round 5, fourth collection: full suite + bench on the centred shadow / tail overlap; prober ablation; f64 VALU rates; centre-only A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05d}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_suite.txt 2>&1; tail -5 $OUT/${TAG}_gpu_suite.txt
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 600 $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
PRAG_LIB=$R/probing-rag_amd/lib/libprag_diag.so timeout 600 python tools/prober_ablate.py > $OUT/${TAG}_prober_ablate.txt 2>&1; cat $OUT/${TAG}_prober_ablate.txt
timeout 120 tools/micro/f64_valu_probe > $OUT/${TAG}_f64_valu_probe.txt 2>&1; cat $OUT/${TAG}_f64_valu_probe.txt
PRAG_SHADOW_AFFINE=2 timeout 300 python tools/embedding_probe.py 1048576 > $OUT/${TAG}_embedding_probe_centre_only.txt 2>/dev/null; python - <<PY
import json
for l in open("$OUT/${TAG}_embedding_probe_centre_only.txt"):
    r = json.loads(l)
    if r["queries"] == 64:
        print("centre only:", r["rows"], r["metric"], r["store"], r["structure"], "two-level %.3f ms fb %d surv %s" % (
            r["two_level"]["ms_per_search"], r["two_level"]["exact_fallbacks_last_search"], r["two_level"].get("survivors")))
PY
