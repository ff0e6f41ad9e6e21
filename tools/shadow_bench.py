#!/usr/bin/env python3
"""Scan time with and without the 8-bit shadow (diagnostic): python tools/shadow_bench.py [docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000
d = 768
for store, metric, k in (("f16", "cos", 10), ("f32", "l2", 5)):
    ix = pra.HipFlatIndex(d, metric, store, capacity=N)
    ix.add_synthetic(42, 0, N)
    Q = torch.from_numpy(synth_rows(7, 0, 64, d)).cuda()
    for B in (64, 32, 1):
        res = {}
        for mode in (0, 1):
            ix.set_shadow(mode)
            for _ in range(3): out = ix.search(Q[:B], k)
            torch.cuda.synchronize()
            ix.profile(256)
            t0 = time.perf_counter()
            for _ in range(20): out = ix.search(Q[:B], k)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20 * 1e3
            kern = np.mean(ix.profile_read()); ix.profile(0)
            res[mode] = (dt, kern, out, ix.last_exact_fallbacks())
        same = torch.equal(res[0][2][1], res[1][2][1])
        print(f"{store} {metric} N={N} B={B:3d} k={k}: plain {res[0][0]:7.3f} ms (kernel {res[0][1]:6.3f}) | shadow {res[1][0]:7.3f} ms "
              f"(kernel {res[1][1]:6.3f}) fallbacks {res[1][3]} ids_equal {same}", flush=True)
    ix.close()
