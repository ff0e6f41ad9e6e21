#!/bin/bash
# GPU-box helper: rocprofv3 passes for the two hot kernels.  usage: tools/gpu_profile.sh <tag> [pmc counters...]
# Runs from /tmp (rocprofv3 writes scratch files to cwd), one --pmc pass per counter (never combined
# with trace domains), then a --kernel-trace --stats pass of bench.py.  Outputs under gpurun_out/<tag>_*.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc/pmc_$c -- python3 $R/tools/prof_kernels.py ${PROF_ARGS:---skip-scan} > $OUT/${TAG}_pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.log
ls $OUT/${TAG}_stats/* | head
