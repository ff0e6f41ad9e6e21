#!/usr/bin/env python3
"""BASELINE config 3 (1k queries x 1M rows x 768, cosine top-10) a few times: the program to put
after `rocprofv3 --kernel-trace --stats --`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N, d, B, k = int(os.environ.get("C3_N", 1_000_000)), 768, int(os.environ.get("C3_B", 1000)), 10
ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
ix.add_synthetic(42, 0, N)
Q = torch.from_numpy(synth_rows(7, 0, B, d)).cuda()
n = int(os.environ.get("C3_REPS", 10))
for shadow in ((0,) if os.environ.get("C3_ONLY16") else (0, 2)):       # 0: fp16 tiles; 2: int8 tiles over the 8-bit shadow first (PRAG_MM8=0 switches them off)
    ix.set_shadow(shadow)
    ix.prepare()
    for _ in range(3):
        ix.search(Q, k)
    torch.cuda.synchronize()
    ix.profile(256)
    t0 = time.perf_counter()
    for _ in range(n):
        D, I = ix.search(Q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    seg = np.asarray(ix.profile_read())
    ix.profile(0)
    print(f"shadow={shadow} {B} x {N} x {d}: {dt*1e3:.3f} ms/search -> {B*N/dt:.3e} scores/s, {2*B*N*d/dt/1e12:.0f} Top/s; "
          f"largest segment {seg.mean():.3f} ms; int8 tier failed {ix.last_tiled8()}, exact fallbacks {ix.last_exact_fallbacks()}",
          flush=True)
    if shadow == 0:
        I_ref = I.clone()
    else:
        print("ids identical to the fp16 tiles:", bool(torch.equal(I, I_ref)), flush=True)
