#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_*/…/*_counter_collection.csv)
into profiles/<tag>_pmc_summary.json and profiles/pmc_scan_topk.json (the file
bench.py reads for roofline.traffic).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide
(16 B/lane) coalesced streaming read, so reads = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact.  Counters were collected in separate --pmc passes.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, src, rows_per_launch, store):
    agg = collections.defaultdict(list)
    rows = []
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        rows += list(csv.DictReader(open(f)))
    # a search launches the scan kernel twice: the 8192-row pre-pass (a few workgroups) and the
    # full pass; only full-grid launches are the "dominant kernel" of the roofline
    full_grid = max((int(r["Grid_Size"]) for r in rows if "scan_topk" in r["Kernel_Name"]), default=0)
    for r in rows:
        k = r["Kernel_Name"]
        name = ("scan_topk_kernel" if "scan_topk" in k else "scan8_kernel" if "scan8" in k else
                "prober16_kernel" if "prober16" in k else "prober_fused_kernel" if "prober_fused" in k else None)
        if name == "scan_topk_kernel" and int(r["Grid_Size"]) != full_grid:
            continue
        if name:
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    out = collections.defaultdict(dict)
    for (k, c), v in agg.items():
        out[k][c] = sum(v) / len(v)
    for k, c in out.items():
        if "FETCH_SIZE" in c:
            c["hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)"] = 2 * c["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in c:
            c["hbm_write_bytes_per_launch(WRITE_SIZE*1024)"] = c["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in c:
            c["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    scan_name = "scan8_kernel" if "scan8_kernel" in out else "scan_topk_kernel"
    s = out.get(scan_name, {})
    if "FETCH_SIZE" in s:
        rec = {"kernel": scan_name, "rows_per_launch": rows_per_launch, "store": store,
               "hbm_bytes_per_launch": s["hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)"] +
                                       s.get("hbm_write_bytes_per_launch(WRITE_SIZE*1024)", 0.0),
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); reads doubled per "
                         "MI355X_MICROARCH.md gfx950 correction", "source": f"profiles/{tag}_pmc_summary.json"}
        json.dump(rec, open(os.path.join(ROOT, "profiles", "pmc_scan_topk.json"), "w"), indent=1)
    print(json.dumps(out, indent=1, sort_keys=True)[:1500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
