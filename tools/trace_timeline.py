#!/usr/bin/env python3
"""Kernel timeline of one steady-state pass out of a rocprofv3 --kernel-trace csv: tools/trace_timeline.py <csv> <name
fragment of the pass's first kernel> [which pass from the end]."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
frag = sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 10
starts = [i for i, n in enumerate(names) if frag in n and (i == 0 or frag not in names[i - 1])]
s, e = starts[-back], starts[-back + 1]
t0 = int(rows[s]['Start_Timestamp']); prev = t0
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].split('(')[0][-52:]
    print(f"{(st - t0) / 1e3:8.1f} us gap {(st - prev) / 1e3:6.1f} dur {(en - st) / 1e3:7.1f} grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:<3} wg {r['Workgroup_Size_X']:>4} {n}")
    prev = en
print("pass", (int(rows[e]['Start_Timestamp']) - t0) / 1e3)
