#!/bin/bash
# Same-box A/B of up to three builds (libprag.so, libprag_ab.so = `make ab ABFLAGS=...`, libprag_prev.so = a kept copy) on the two-level scan
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-ab}; export SCAN8_AB_Q=${2:-64,128}
rm -f $OUT/${TAG}_ab.txt
for rep in 1 2 3; do
  for lib in libprag.so libprag_ab.so libprag_prev.so; do
    [ -f $R/probing-rag_amd/lib/$lib ] || continue
    PRAG_LIB=$R/probing-rag_amd/lib/$lib timeout 300 python tools/scan8_ab.py 21000000 40 2>&1 | tail -1 >> $OUT/${TAG}_ab.txt
  done
done
cat $OUT/${TAG}_ab.txt
