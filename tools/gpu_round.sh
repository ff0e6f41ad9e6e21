#!/bin/bash
# (each *_bench*.json holds two lines: `BENCH_DETAIL {full record}` and the compact line the driver parses - the LAST line)
# Collection of a round's GPU record: full suite, bench (default = fused pass), shard-size bench, 2-rank gloo bench,
# e2e loop, rocprof kernel stats of the bench command, fuzz soaks.  usage: gpu_round.sh <tag> [fuzz seconds scale]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r06}; FS=${2:-1}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_suite.txt 2>&1; tail -4 $OUT/${TAG}_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 300 $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
timeout 300 python bench.py --docs 2625000 --gate-batch 512 --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_shard_2625000_gate512.json 2>> $OUT/${TAG}_bench.err
PRAG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_2ranks_gloo.json 2> $OUT/${TAG}_bench_2ranks_gloo.err; echo "gloo rc=$?"; tail -2 $OUT/${TAG}_bench_2ranks_gloo.err
timeout 600 python bench.py --e2e --e2e-queries 200 --no-cpu-baseline > $OUT/${TAG}_e2e_200q.json 2> $OUT/${TAG}_e2e.err
python - <<PY
import json
for n in ("bench", "bench_shard_2625000_gate512", "bench_2ranks_gloo"):
    try:
        r = json.loads(open("$OUT/${TAG}_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "ms_per_pass", r["config"]["ms_per_pass"], "scan", r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], "gate", (r["roofline"].get("gate") or {}).get("avg_launch_ms"),
              "decisions/s", r.get("probe_decisions_per_s"), "plan", r.get("plan"), "exchange", r.get("exchange"), "shard", r["config"].get("shard_pass_ms"),
              r["config"].get("shard_pass_mode"), "q128", r["config"].get("q128_shadow_ms"), "c3", r["config"].get("c3_ms"))
    except Exception as e:
        print(n, "unreadable:", e)
try:
    r = json.loads(open("$OUT/${TAG}_e2e_200q.json").read().strip().splitlines()[-1])
    print("e2e", r["value"], r["hip_path"].get("us_per_call"), r["hip_path"].get("clock_overhead_us_per_call"))
except Exception as e:
    print("e2e unreadable:", e)
PY
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.log)
f=$(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1); cp $f $OUT/${TAG}_bench_kernel_stats.csv; head -8 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-170; rm -rf $OUT/${TAG}_stats
timeout $((300 * FS + 30)) python tools/fuzz_shadow.py $((300 * FS)) 5 > $OUT/${TAG}_fuzz_shadow.txt 2>&1; tail -2 $OUT/${TAG}_fuzz_shadow.txt
timeout $((170 * FS + 30)) python tools/fuzz_paths.py $((170 * FS)) 5 > $OUT/${TAG}_fuzz_paths.txt 2>&1; tail -2 $OUT/${TAG}_fuzz_paths.txt
timeout $((120 * FS + 30)) python tools/fuzz_prober.py $((120 * FS)) 5 > $OUT/${TAG}_fuzz_prober.txt 2>&1; tail -2 $OUT/${TAG}_fuzz_prober.txt
