mkdir -p gpurun_out
bash tools/gpu_profile.sh r04q > gpurun_out/r04q_profile.log 2>&1
tail -3 gpurun_out/r04q_profile.log
timeout 1200 python bench.py > gpurun_out/r04q_bench.json 2> gpurun_out/r04q_bench.err; tail -c 300 gpurun_out/r04q_bench.err
python - <<PY
import json
r=json.loads(open("gpurun_out/r04q_bench.json").read().strip().splitlines()[-1])
print({k:v for k,v in r.items() if k not in ("variants","config")})
print(r["config"])
PY
