#!/usr/bin/env python3
"""Prober C2 timing (6 layers, B=4096, d=2048, f16): kernel time back to back and after a cache flush.
Argument: d_model (the slope over d separates the fc1 loop from the fixed part).  Diagnostic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import random_prober_state

L, B = 6, 4096
D = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ens = pra.HipProberEnsemble(L, D, 2, weights="f16")
for l in range(L):
    ens.load_layer(l, random_prober_state(100 + l, D))
x = torch.randn(L, B, D, device="cuda").half()
ref = ens.forward(x).float().clone() if hasattr(ens, "forward") else None
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for mode in ("back-to-back", "flushed"):
    ens.profile(256)
    for _ in range(100):
        if mode == "flushed":
            flush.add_(1)
        ens.gate(x, 0, 0.0)
    torch.cuda.synchronize()
    t = np.asarray(ens.profile_read()) * 1e3
    ens.profile(0)
    print(f"d={D} {mode:13s}: median {np.median(t):6.1f} us  min {t.min():6.1f}  mean {t.mean():6.1f}", flush=True)
out = ens.forward(x).float() if ref is not None else None
if ref is not None:
    print("checksum", float(out.double().sum()), "repeatable", bool(torch.equal(out, ref)))
