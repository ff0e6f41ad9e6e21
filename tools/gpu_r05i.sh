#!/bin/bash
# round 5, ninth collection: gate folded into the prober launch, adaptive gather launch, fused pass: tests + shard A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05i}
timeout 1200 python -m pytest tests/test_gpu_prober.py tests/test_gpu_shadow.py tests/test_gpu_tiled8.py tests/test_gpu_loop.py tests/test_gpu_configs.py -x -q > $OUT/${TAG}_tests.txt 2>&1; tail -4 $OUT/${TAG}_tests.txt
rm -f $OUT/${TAG}_shard_ab.txt
for sw in "" "SHARD_FUSED=1" "PRAG_GATHER=1 PRAG_GATE_FOLD=0" "PRAG_GATHER=1 PRAG_GATE_FOLD=0 SHARD_FUSED=1" "SHARD_TAIL=1" "" "SHARD_FUSED=1"; do
  echo "== $sw" >> $OUT/${TAG}_shard_ab.txt
  env $sw SHARD_REPS=400 timeout 200 python tools/shard_pass.py 2>&1 | grep "shard pass" | cut -c1-60 >> $OUT/${TAG}_shard_ab.txt
done
cat $OUT/${TAG}_shard_ab.txt
cd /tmp && export TMPDIR=/tmp
SHARD_FUSED=1 SHARD_REPS=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_fused_stats -- python3 $R/tools/shard_pass.py > $OUT/${TAG}_fused_pass_under_rocprof.txt 2>&1
f=$(ls $OUT/${TAG}_fused_stats/*/*kernel_stats.csv | head -1); cp $f $OUT/${TAG}_fused_pass_kernel_stats.csv; head -10 $OUT/${TAG}_fused_pass_kernel_stats.csv | cut -c1-150
rm -rf $OUT/${TAG}_fused_stats
