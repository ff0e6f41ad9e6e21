#!/usr/bin/env python3
"""Prober kernel time over the batch sizes a rank sees when the gate rows are split across 1..8 GPUs
(6 layers, d=2048): back to back and after a cache flush.  Arguments: weights mode (f16|f32), x dtype (f16|f32).
PRAG_PROBER_WD=1 keeps one K step of weights in flight in every tile shape (A/B).  Diagnostic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import random_prober_state

L, D = 6, 2048
wmode = sys.argv[1] if len(sys.argv) > 1 else "f16"
xdt = sys.argv[2] if len(sys.argv) > 2 else "f16"
ens = pra.HipProberEnsemble(L, D, 2, weights=wmode)
for l in range(L):
    ens.load_layer(l, random_prober_state(100 + l, D))
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
print(f"weights {wmode}, x {xdt}, PRAG_PROBER_WD={os.environ.get('PRAG_PROBER_WD', '(default)')}")
for B in (8, 32, 128, 256, 512, 1024, 1365, 2048, 4096):
    x = torch.randn(L, B, D, device="cuda")
    if xdt == "f16":
        x = x.half()
    line = f"B={B:5d}:"
    for mode in ("back-to-back", "flushed"):
        ens.profile(256)
        for _ in range(60):
            if mode == "flushed":
                flush.add_(1)
            ens.gate(x, 0, 0.0)
        torch.cuda.synchronize()
        t = np.asarray(ens.profile_read()) * 1e3
        ens.profile(0)
        line += f"  {mode} median {np.median(t):6.1f} us (min {t.min():6.1f})"
    lg = ens.forward(x).float()
    line += f"  checksum {float(lg.double().sum()):.6f}"
    print(line, flush=True)
