mkdir -p gpurun_out; R=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04p_gpu_suite.txt 2>&1; tail -3 gpurun_out/r04p_gpu_suite.txt
PRAG_LIB=$PWD/probing-rag_amd/lib/libprag_diag.so PRAG_SHADOW_DBG=512 python tools/gather_probe.py child 2>&1 | grep "bound" | sort | uniq -c | sort -k1nr | head -4
for m in 0 1; do PRAG_SHADOW_BOUND=$m python bench.py --no-cpu-baseline > gpurun_out/r04p_bench_bound$m.json 2>/dev/null; done
python - <<PY
import json
for m in (0,1):
    r=json.loads(open(f"gpurun_out/r04p_bench_bound{m}.json").read().strip().splitlines()[-1])
    print("bound",m,"pass", r["config"]["ms_per_pass"], "scan", r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], "gate", r["roofline_gate"]["avg_launch_ms"])
    for k,v in r["variants"].items():
        if isinstance(v,dict): print("   ",k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_search","frac","pass_ms","search_alone_ms","predicted_strong_scaling_eff","us_per_decision","error")})
PY
