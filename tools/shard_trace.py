#!/usr/bin/env python3
"""Per-kernel medians of a search at one shard size from a rocprofv3 --kernel-trace CSV (diagnostic):
python tools/shard_trace.py <dir with *_kernel_trace.csv>"""
import collections, csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
out = collections.defaultdict(list)
for i, (n, s, e) in enumerate(seq):
    if "scan8_kernel" in n and 0 < i and i + 2 < len(seq):
        qt = "64-query tiles" if "<64" in n else "32-query tiles"
        prep, g, ex = seq[i - 1], seq[i + 1], seq[i + 2]
        if "prep_queries" in prep[0] and "gather" in g[0]:
            out[qt].append(((prep[2] - prep[1]) / 1e3, (s - prep[2]) / 1e3, (e - s) / 1e3, (g[1] - e) / 1e3,
                            (g[2] - g[1]) / 1e3, (ex[1] - g[2]) / 1e3, (ex[2] - ex[1]) / 1e3))
for k, v in out.items():
    a = np.median(np.array(v), axis=0)
    print(f"{k} ({len(v)} searches): prep {a[0]:.1f} | gap {a[1]:.1f} | scan8 {a[2]:.1f} | gap {a[3]:.1f} | gather {a[4]:.1f} | gap {a[5]:.1f} | "
          f"exact {a[6]:.1f} | sum {a.sum():.1f} us")
