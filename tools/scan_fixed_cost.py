#!/usr/bin/env python3
"""Fixed cost of a scan_topk launch: kernel time vs rows per workgroup (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
d, k = 768, 10
for B in (64, 32):
    Q = torch.from_numpy(synth_rows(7, 0, B, d)).cuda()
    for N in (256, 8192, 65536, 100_000, 262_144, 2_625_000):
        ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
        ix.add_synthetic(42, 0, N)
        ix.profile(256)
        for _ in range(30): ix.search(Q, k)
        torch.cuda.synchronize()
        t = np.array(ix.profile_read())[5:]
        print(f"B={B} N={N:8d}: scan kernel median {np.median(t)*1e3:7.1f} us (ideal at 6.2 TB/s {N*d*2/6.2e12*1e6:6.1f})", flush=True)
        del ix
