#!/bin/bash
# round 5: same-box A/B of libprag.so against libprag_ab.so (a compile-time alternative, see csrc/Makefile `ab`)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05m}; TESTS=${2:-tests/test_gpu_shadow.py}
timeout 900 python -m pytest $TESTS -m gpu -x -q > $OUT/${TAG}_tests.txt 2>&1; tail -3 $OUT/${TAG}_tests.txt
AB=$R/probing-rag_amd/lib/libprag_ab.so
rm -f $OUT/${TAG}_ab.txt
for rep in 1 2 3; do
  for lib in new ab; do
    if [ $lib = ab ]; then export PRAG_LIB=$AB; else unset PRAG_LIB; fi
    timeout 300 python tools/scan8_ab.py 21000000 40 2>&1 | tail -1 >> $OUT/${TAG}_ab.txt
  done
done
for lib in new ab; do
  if [ $lib = ab ]; then export PRAG_LIB=$AB; else unset PRAG_LIB; fi
  timeout 300 python tools/scan8_ab.py 2625000 200 2>&1 | tail -1 >> $OUT/${TAG}_ab.txt
done
unset PRAG_LIB
cat $OUT/${TAG}_ab.txt
