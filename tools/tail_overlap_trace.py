#!/usr/bin/env python3
"""Timeline of the shard pass with the gate beside the search's tail, from a rocprofv3 --kernel-trace CSV
(tools/shard_pass.py with SHARD_TAIL=1): for every pass, when the gate's prober kernel starts and ends relative to
the end of scan8 and to the bound kernel - how much of the overlap the cross-stream dependency leaves.
python tools/tail_overlap_trace.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
scans = [i for i, e in enumerate(ev) if "scan8_kernel" in e[0]]
out = []
for a, b in zip(scans[20:-2], scans[21:-1]):
    s_end = ev[a][2]
    seg = ev[a + 1:b + 1]
    def first(name):
        for n, s, e in seg:
            if name in n:
                return s, e
        return None
    bound, prob, gate, gather, exact, prep = first("shadow_bound"), first("prober16"), first("gate_kernel"), first("shadow_gather"), first("exact_scan"), first("prep_queries")
    nxt = ev[b][1]
    if not (bound and prob and prep):
        continue
    out.append(((bound[0] - s_end) / 1e3, (bound[1] - bound[0]) / 1e3, (prob[0] - s_end) / 1e3, (prob[1] - prob[0]) / 1e3,
                ((gate[1] if gate else prob[1]) - s_end) / 1e3, ((exact[1] if exact else bound[1]) - s_end) / 1e3,
                (prep[0] - s_end) / 1e3, (prep[1] - prep[0]) / 1e3, (nxt - s_end) / 1e3, (ev[b][2] - ev[b][1]) / 1e3))
a = np.median(np.array(out), axis=0)
print(f"{len(out)} passes, medians in us, t = 0 at the end of scan8:")
print(f"  bound kernel starts at {a[0]:.1f}, runs {a[1]:.1f}; search tail (bound + gather + exact probe) ends at {a[5]:.1f}")
print(f"  prober16 (side stream) starts at {a[2]:.1f}, runs {a[3]:.1f}; gate done at {a[4]:.1f}")
print(f"  next pass: prep starts at {a[6]:.1f}, runs {a[7]:.1f}; next scan8 starts at {a[8]:.1f} and runs {a[9]:.1f}")
