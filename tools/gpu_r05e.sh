#!/bin/bash
# round 5, fifth collection: centre-only default + retry tier: shadow / index / config tests, embedding probe at 4 M
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05e}
timeout 1500 python -m pytest tests/test_gpu_shadow.py tests/test_gpu_tiled8.py tests/test_gpu_index.py tests/test_gpu_configs.py -x -q > $OUT/${TAG}_tests.txt 2>&1; tail -5 $OUT/${TAG}_tests.txt
timeout 400 python tools/embedding_probe.py 4194304 > $OUT/${TAG}_embedding_probe.txt 2> $OUT/${TAG}_embedding_probe.err; python - <<PY
import json
for l in open("$OUT/${TAG}_embedding_probe.txt"):
    r = json.loads(l)
    print(r["rows"], r["metric"], r["store"], r["queries"], r["structure"], "direct %.3f (fb %d) two-level %.3f ms fb %d surv %s" % (
        r["rows_scanned_directly"]["ms_per_search"], r["rows_scanned_directly"]["exact_fallbacks_last_search"], r["two_level"]["ms_per_search"], r["two_level"]["exact_fallbacks_last_search"],
        r["two_level"].get("survivors")))
PY
