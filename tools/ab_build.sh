#!/bin/bash
# Build probing-rag_amd/lib/libprag_prev.so from the committed (HEAD) version of ONE csrc file plus the
# current objects of the others: same-box A/B of a kernel change (`PRAG_LIB=.../libprag_prev.so`).
# usage: tools/ab_build.sh prober.hip     (run `make -C probing-rag_amd/csrc` first)
set -e
F=$1
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/probing-rag_amd/csrc
git show HEAD:probing-rag_amd/csrc/$F > _prev_$F
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -c _prev_$F -o ../lib/obj/_prev.o
rm _prev_$F
OBJS=""
for s in common prober prober_small flat_index flat_mm flat_exact flat_shadow trainer; do
  if [ "$s.hip" = "$F" ]; then OBJS="$OBJS ../lib/obj/_prev.o"; else OBJS="$OBJS ../lib/obj/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/libprag_prev.so $OBJS
ls -la ../lib/libprag_prev.so
