#!/usr/bin/env python3
"""Query-stationary scan (65..128 queries): parity on small shards and speed at 128 x 21M (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from oracle import oracle_np as onp
for metric, mid in (("l2", onp.METRIC_L2), ("cos", onp.METRIC_COS)):
    N, B, d, k = 300_000, 128, 768, 10
    ix = pra.HipFlatIndex(d, metric, "f16", capacity=N)
    ix.add_synthetic(11, 0, N)
    Q = onp.synth_rows(12, 0, B, d)
    D, I = ix.search(Q, k)
    from oracle import oracle_c
    D0, I0 = oracle_c.flat_search(ix.reconstruct_n(0, N), Q[:16], k, mid)
    print(metric, "ids exact:", np.array_equal(I[:16], I0), flush=True)
    del ix
for metric in ("cos", "l2"):
    N, d, B, k = int(os.environ.get("QS_N", 21_000_000)), 768, 128, 10
    ix = pra.HipFlatIndex(d, metric, "f16", capacity=N)
    ix.add_synthetic(42, 0, N)
    Q = torch.from_numpy(onp.synth_rows(7, 0, B, d)).cuda()
    ix.profile(64)
    for _ in range(2): ix.search(Q, k)
    torch.cuda.synchronize(); ix.profile_read()
    t0 = time.perf_counter()
    for _ in range(5): ix.search(Q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    ker = np.mean(ix.profile_read())
    print(f"{metric}: 128 x {N}: search {dt*1e3:.3f} ms, scan kernel {ker:.3f} ms = {N*d*2/ker/1e9:.2f} TB/s -> {B*N/dt:.3e} scores/s", flush=True)
    del ix
