#!/bin/bash
# build probing-rag_amd/lib/libprag_<tag>.so with extra -D flags for ONE source file (experiments): tools/build_variant.sh <tag> <file.hip> <flags...>
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../probing-rag_amd/csrc"
mkdir -p ../lib/obj_$TAG
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I../../include -I. "$@" -c $SRC -o ../lib/obj_$TAG/${SRC%.hip}.o
OBJS=""
for o in ../lib/obj/*.o; do b=$(basename $o); if [ "$b" == "${SRC%.hip}.o" ]; then OBJS="$OBJS ../lib/obj_$TAG/$b"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/libprag_$TAG.so $OBJS -ldl
echo built ../lib/libprag_$TAG.so
