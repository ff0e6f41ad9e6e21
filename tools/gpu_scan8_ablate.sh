#!/bin/bash
# round 5: timing-only ablations of scan8 on the fragment-major shadow, 128 and 64 queries (make diag build; results WRONG)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-ablate}
for nq in 128 64 32; do
  echo "== $nq queries" >> $OUT/${TAG}_scan8_ablation.txt
  PRAG_QUERIES=$nq ABLATE=0,2,1024,2048,3072,4096,16384,17408,32768 timeout 900 python tools/scan8_ablate.py 2>&1 | grep PRAG_SHADOW_DBG >> $OUT/${TAG}_scan8_ablation.txt
done
cat $OUT/${TAG}_scan8_ablation.txt
