#!/usr/bin/env python3
"""65..128 queries: one pass over the 8-bit shadow (128-query tiles) vs the query-stationary fp16 scan (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000
d, k = 768, 10
ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
ix.add_synthetic(42, 0, N)
Q = torch.from_numpy(synth_rows(7, 0, 128, d)).cuda()
res = {}
for B in (128, 96, 65):
    for mode in (0, 1):
        ix.set_shadow(mode); ix.prepare()
        for _ in range(3): out = ix.search(Q[:B], k)
        torch.cuda.synchronize()
        ix.profile(256)
        t0 = time.perf_counter()
        for _ in range(20): out = ix.search(Q[:B], k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        kern = float(np.mean(ix.profile_read())); ix.profile(0)
        res[mode] = (out, dt, kern, ix.last_exact_fallbacks())
    same = torch.equal(res[0][0][1], res[1][0][1])
    print(f"N={N} B={B}: fp16 rows {res[0][1]:.3f} ms (kernel {res[0][2]:.3f}) | two-level {res[1][1]:.3f} ms (kernel {res[1][2]:.3f}, "
          f"{N * (d + 12) / res[1][2] / 1e6 / 8000:.3f} of 8 TB/s) fallbacks {res[1][3]} ids_equal {same}", flush=True)
