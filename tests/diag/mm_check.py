#!/usr/bin/env python3
"""MFMA-tiled scan (> 128 queries): parity against the oracle and the per-lane-list kernels, timing (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from oracle import oracle_np as onp

ok = True
for d in (256, 768):
    for metric, mid in (("l2", onp.METRIC_L2), ("ip", onp.METRIC_IP), ("cos", onp.METRIC_COS)):
        for N, B, k in ((20_000, 300, 5), (3_000, 129, 10), (70_000, 513, 10)):
            ix = pra.HipFlatIndex(d, metric, "f16")
            ix.add_synthetic(11, 0, N)
            X = ix.reconstruct_n(0, N)
            Q = onp.synth_rows(12, 0, B, d)
            D, I = ix.search(Q, k)
            D0, I0 = onp.flat_search(X, Q, k, mid)
            same = np.array_equal(I, I0)
            err = np.abs(D - D0).max() / max(1.0, np.abs(D0).max())
            print(f"d={d} {metric} N={N} B={B} k={k}: ids {'OK' if same else 'DIFF'} ({(I != I0).sum()} mismatches) rel err {err:.2e}", flush=True)
            ok &= same
            del ix
print("PARITY", "OK" if ok else "FAILED", flush=True)

# BASELINE config 3: 1k x 1M x 768 cosine top-10
N, d, B, k = 1_000_000, 768, 1000, 10
ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
ix.add_synthetic(42, 0, N)
Q = torch.from_numpy(onp.synth_rows(7, 0, B, d)).cuda()
D, I = ix.search(Q, k)
torch.cuda.synchronize()
os.environ["PRAG_SCAN_MM"] = "0"
ix2 = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
ix2.add_synthetic(42, 0, N)
D2, I2 = ix2.search(Q, k)
torch.cuda.synchronize()
print("C3 ids equal to the list kernels:", bool((I == I2).all()), "scores equal:", bool((D == D2).all()), flush=True)
for name, x in (("mm", ix), ("lists", ix2)):
    for _ in range(3):
        x.search(Q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        x.search(Q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"C3 {name}: {dt*1e3:.3f} ms -> {B*N/dt:.3e} scores/s, {2*B*N*d/dt/1e12:.1f} TFLOP/s ({2*B*N*d/dt/2.5e15:.3f} of 2.5 PF)", flush=True)
