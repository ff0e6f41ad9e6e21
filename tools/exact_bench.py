#!/usr/bin/env python3
"""Speed of the exact float64 fallback scan (diagnostic): k = 100 on d = 640 goes straight to it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000, 640
for store in ("f16", "f32"):
    ix = pra.HipFlatIndex(d, "l2", store, capacity=N)
    ix.add_synthetic(42, 0, N)
    q = torch.from_numpy(synth_rows(7, 0, 4, d)).cuda()
    for B in (1, 4):
        for _ in range(2): ix.search(q[:B], 100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): ix.search(q[:B], 100)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        gb = N * d * (2 if store == "f16" else 4) * B / 1e9
        print(f"{store} N={N} B={B}: {dt*1e3:8.2f} ms per search = {gb/dt/1e3:5.2f} TB/s over {B} pass(es), fallbacks {ix.last_exact_fallbacks()}", flush=True)
    ix.close()
