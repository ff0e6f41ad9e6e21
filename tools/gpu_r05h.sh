#!/bin/bash
# round 5, eighth collection: the fused pass (tests + A/B at the shard size), tiled8 / index test files, exact fallback A/B, e2e loop,
# rocprof kernel stats of the bench command and of the shard pass, timeline of the tail overlap
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r05h}
timeout 900 python -m pytest tests/test_gpu_shadow.py tests/test_gpu_tiled8.py tests/test_gpu_index.py tests/test_gpu_prober.py -x -q > $OUT/${TAG}_tests.txt 2>&1; tail -4 $OUT/${TAG}_tests.txt
rm -f $OUT/${TAG}_shard_ab.txt
for sw in "" "SHARD_TAIL=1" "SHARD_FUSED=1" "SHARD_TAIL=1" "SHARD_FUSED=1" "PRAG_SHADOW_SAMPLE=3 SHARD_FUSED=1" "PRAG_SHADOW_SAMPLE=0 SHARD_FUSED=1"; do
  echo "== $sw" >> $OUT/${TAG}_shard_ab.txt
  env $sw SHARD_REPS=400 timeout 200 python tools/shard_pass.py 2>&1 | grep "shard pass" | cut -c1-60 >> $OUT/${TAG}_shard_ab.txt
done
cat $OUT/${TAG}_shard_ab.txt
timeout 400 python tools/exact_group_bench.py > $OUT/${TAG}_exact_group_bench.txt 2>&1; cat $OUT/${TAG}_exact_group_bench.txt
timeout 600 python bench.py --e2e --e2e-queries 200 --no-cpu-baseline > $OUT/${TAG}_e2e_200q.json 2> $OUT/${TAG}_e2e.err; tail -c 1000 $OUT/${TAG}_e2e_200q.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.log
f=$(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1); cp $f $OUT/${TAG}_bench_kernel_stats.csv; head -8 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-160
SHARD_TAIL=1 SHARD_REPS=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_shard_stats -- python3 $R/tools/shard_pass.py > $OUT/${TAG}_shard_pass_under_rocprof.txt 2>&1
python3 $R/tools/tail_overlap_trace.py $OUT/${TAG}_shard_stats > $OUT/${TAG}_tail_overlap_trace.txt 2>&1; cat $OUT/${TAG}_tail_overlap_trace.txt
SHARD_FUSED=1 SHARD_REPS=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_fused_stats -- python3 $R/tools/shard_pass.py > $OUT/${TAG}_fused_pass_under_rocprof.txt 2>&1
f=$(ls $OUT/${TAG}_fused_stats/*/*kernel_stats.csv | head -1); cp $f $OUT/${TAG}_fused_pass_kernel_stats.csv; head -12 $OUT/${TAG}_fused_pass_kernel_stats.csv | cut -c1-160
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_shard_stats $OUT/${TAG}_fused_stats
