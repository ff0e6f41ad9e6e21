#!/usr/bin/env python3
"""Searches with k > 26 (deep candidate lists through the MFMA-tiled scan): time at 21M rows (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N, d = int(os.environ.get("LK_N", 21_000_000)), 768
ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
ix.add_synthetic(42, 0, N)
for B, k in ((1, 10), (1, 100), (1, 900), (64, 100), (256, 100)):
    Q = torch.from_numpy(synth_rows(7, 0, B, d)).cuda()
    for _ in range(2): ix.search(Q, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): D, I = ix.search(Q, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    ok = bool((torch.diff(D, dim=1) <= 0).all()) and len(set(I[0].tolist())) == k
    print(f"B={B:4d} k={k:4d}: {dt*1e3:8.3f} ms/search (sorted, unique: {ok})", flush=True)
