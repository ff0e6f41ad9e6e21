#!/usr/bin/env python3
"""Fused prober, fp16 x fp16 mode: 16 x 16 MFMA tiles (prober16.hip) against the 32 x 32 kernel (PRAG_PROBER_SHAPE=32),
two handles in ONE process, interleaved rounds, back to back and behind a cache flush."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import random_prober_state

L, D = 6, 2048
def make(shape):
    if shape == 32: os.environ["PRAG_PROBER_SHAPE"] = "32"
    else: os.environ.pop("PRAG_PROBER_SHAPE", None)
    e = pra.HipProberEnsemble(L, D, 2, weights="f16")
    os.environ.pop("PRAG_PROBER_SHAPE", None)
    for l in range(L): e.load_layer(l, random_prober_state(100 + l, D))
    return e
ens = {32: make(32), 16: make(16)}
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for B in (512, 1024, 2048, 4096):
    x = torch.randn(L, B, D, device="cuda").half()
    out = (torch.empty((L, B, 2), device="cuda"), torch.empty((B, 2), device="cuda"), torch.empty((B,), dtype=torch.int32, device="cuda"))
    res = {}
    for e in ens.values():
        for _ in range(10): e.gate(x, 0, 0.0, out=out)
    torch.cuda.synchronize()
    t = {32: [], 16: []}; tf = {32: [], 16: []}
    for rnd in range(9):
        for shape in ((32, 16) if rnd % 2 == 0 else (16, 32)):
            e = ens[shape]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): e.gate(x, 0, 0.0, out=out)
            torch.cuda.synchronize(); t[shape].append((time.perf_counter() - t0) / 50 * 1e6)
            e.profile(16)
            for _ in range(8):
                flush.add_(1)
                e.gate(x, 0, 0.0, out=out)
            torch.cuda.synchronize()
            tf[shape].append(float(np.median(e.profile_read())) * 1e3); e.profile(0)
            res[shape] = out[0].clone()
    diff = float((res[32] - res[16]).abs().max())
    print(f"B={B:5d}: back to back 32x32 {np.median(t[32]):7.1f} us | 16x16 {np.median(t[16]):7.1f} us | ratio {np.median(t[16]) / np.median(t[32]):.3f} || "
          f"behind a 1 GiB flush (kernel events) 32x32 {np.median(tf[32]):7.1f} | 16x16 {np.median(tf[16]):7.1f} | ratio {np.median(tf[16]) / np.median(tf[32]):.3f} || max |dlogit| {diff:.2e}", flush=True)
