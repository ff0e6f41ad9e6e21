#!/usr/bin/env python3
"""BASELINE config 3 (1k queries x 1M rows x 768, cosine top-10) a few times: the program to put
after `rocprofv3 --kernel-trace --stats --`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from oracle import oracle_np as onp
N, d, B, k = int(os.environ.get("C3_N", 1_000_000)), 768, int(os.environ.get("C3_B", 1000)), 10
ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
ix.add_synthetic(42, 0, N)
Q = torch.from_numpy(onp.synth_rows(7, 0, B, d)).cuda()
for _ in range(3):
    ix.search(Q, k)
torch.cuda.synchronize()
n = int(os.environ.get("C3_REPS", 10))
t0 = time.perf_counter()
for _ in range(n):
    ix.search(Q, k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{B} x {N} x {d}: {dt*1e3:.3f} ms/search -> {B*N/dt:.3e} scores/s, {2*B*N*d/dt/1e12:.0f} TFLOP/s", flush=True)
