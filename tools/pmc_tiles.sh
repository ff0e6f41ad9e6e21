#!/bin/bash
# GPU-box helper: --pmc passes (one counter per pass, never combined with trace domains) over the MFMA-tiled
# scans - 1000 queries x C3_N rows, fp16 tiles then int8 tiles (tools/c3_search.py).  usage: tools/pmc_tiles.sh <tag> [rows]
set -u
TAG=$1; export C3_N=${2:-21000000}; export C3_REPS=2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU FETCH_SIZE SQ_LDS_BANK_CONFLICT; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc/pmc_$c -- python3 $R/tools/c3_search.py > $OUT/${TAG}_pmc_$c.log 2>&1
done
python3 - <<PY
import collections, csv, glob, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/${TAG}_pmc/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "scan_mm_kernel" not in k:
            continue
        name = "scan_mm_kernel<int8>" if ", true>" in k else "scan_mm_kernel<fp16>"
        agg[(name, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
out = collections.defaultdict(dict)
for (name, grid, c), v in agg.items():
    out[name].setdefault("sum_over_the_launches_of_one_search", collections.defaultdict(float))
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for (name, grid, c), v in agg.items():
    per[name][c] += sum(v)
    cnt[name][c] += len(v)
res = {}
for name in per:
    d = {c: per[name][c] for c in per[name]}
    d["launches_counted"] = max(cnt[name].values())
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CU_CYCLES" in d:
        d["mfma_busy_frac_of_cu_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * d["SQ_BUSY_CU_CYCLES"])
    if "SQ_INSTS_VALU" in d and "SQ_INSTS_MFMA" in d:
        d["valu_per_mfma"] = d["SQ_INSTS_VALU"] / d["SQ_INSTS_MFMA"]
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes(2*FETCH_SIZE*1024)"] = 2 * d["FETCH_SIZE"] * 1024
    res[name] = d
res["what"] = "counters summed over every scan_mm launch of the run (all segments; $C3_REPS + 3 warm-up searches per mode), 1000 queries x $C3_N rows x 768"
json.dump(res, open("$OUT/${TAG}_pmc_tiles.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True)[:2500])
PY
