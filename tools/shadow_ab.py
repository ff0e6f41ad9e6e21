#!/usr/bin/env python3
"""Two-level search timing at the headline and the 8-GPU-shard sizes (diagnostic):
python tools/shadow_ab.py [tag]  -> one line per (rows, queries): search ms, scan8 kernel ms, fraction of 8 TB/s on
the shadow's bytes, fallbacks; plus the time of add_synthetic (which now includes the shadow build)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
tag = sys.argv[1] if len(sys.argv) > 1 else ""
sizes = [int(x) for x in os.environ.get("AB_SIZES", "21000000,2625000").split(",")]
d, k = 768, 10
Q = torch.from_numpy(synth_rows(7, 0, 64, d)).cuda()
for N in sizes:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
    ix.add_synthetic(42, 0, N)
    torch.cuda.synchronize(); t_add = time.perf_counter() - t0
    t0 = time.perf_counter(); ix.search(Q[:1], k); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
    for B in (64, 32, 1):
        for _ in range(3): ix.search(Q[:B], k)
        torch.cuda.synchronize()
        reps = 200 if N < 5_000_000 else 40
        ix.profile(512)
        t0 = time.perf_counter()
        for _ in range(reps): out = ix.search(Q[:B], k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        kern = float(np.median(ix.profile_read())); ix.profile(0)
        print(f"{tag} N={N} B={B:3d}: search {dt:7.4f} ms, scan8 {kern:7.4f} ms = {N * (d + 12) / kern / 1e6 / 8000:.3f} of 8 TB/s, "
              f"fallbacks {ix.last_exact_fallbacks()}, add+build {t_add*1e3:.0f} ms, first search {t_first*1e3:.2f} ms", flush=True)
    ix.close()
