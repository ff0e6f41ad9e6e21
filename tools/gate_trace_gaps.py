#!/usr/bin/env python3
"""Out of a `rocprofv3 --kernel-trace` csv of tools/gate_in_loop.py (or bench.py --e2e): every B <= 8 gate
(small_fc_kernel x 2 + small_head_kernel, or the one-launch form), its kernels' own durations and the gaps between
them - medians over the trace, the first three gates skipped.   tools/gate_trace_gaps.py <kernel_trace.csv>"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
heads = [i for i, n in enumerate(names) if "small_head_kernel" in n or "small_gate1_kernel" in n]
rec = {}


def add(k, v):
    rec.setdefault(k, []).append(v / 1e3)


for i in heads[3:]:
    j = i
    chain = [i]
    while j > 0 and len(chain) < 4 and ("small_" in names[j - 1] or "pool_accumulate" in names[j - 1] or "prefetch" in names[j - 1]):
        j -= 1
        chain.insert(0, j)
    prev_end = int(rows[chain[0] - 1]["End_Timestamp"]) if chain[0] > 0 else None
    for c in chain:
        st, en = int(rows[c]["Start_Timestamp"]), int(rows[c]["End_Timestamp"])
        nm = names[c].split("(")[0].split("<")[0].split("::")[-1]
        pos = chain.index(c)
        add(f"{pos} {nm} dur", en - st)
        if prev_end is not None and pos > 0:
            add(f"{pos} {nm} gap before", st - prev_end)
        prev_end = en
    add("first kernel start -> head end", int(rows[i]["End_Timestamp"]) - int(rows[chain[0]]["Start_Timestamp"]))
print(f"{len(heads) - 3} gates")
for k in sorted(rec):
    v = np.asarray(rec[k])
    print(f"{k:48s} median {np.median(v):7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}")
