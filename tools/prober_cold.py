#!/usr/bin/env python3
"""Which stream makes the cold prober launch slow?  Kernel time of prober_fused (C2 shape) after a cache
flush, with (a) nothing re-warmed, (b) the activations re-read first, (c) the weights re-read first
(a launch on a second activation tensor).  Diagnostic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import random_prober_state

L, B, D = 6, 4096, 2048
ens = pra.HipProberEnsemble(L, D, 2, weights="f16")
for l in range(L):
    ens.load_layer(l, random_prober_state(100 + l, D))
x = torch.randn(L, B, D, device="cuda").half()
x2 = torch.randn(L, B, D, device="cuda").half()
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
dirty = os.environ.get("PRAG_DIRTY_FLUSH", "0") == "1"


def do_flush():
    if dirty:
        flush.add_(1)
    else:
        flush.view(torch.int32).sum()


for mode in ("warm", "cold", "x re-read", "weights re-read"):
    ens.profile(0)
    times = []
    for _ in range(40):
        if mode != "warm":
            do_flush()
        if mode == "x re-read":
            x.view(torch.int32).sum()
        if mode == "weights re-read":
            ens.gate(x2, 0, 0.0)
        ens.profile(8)
        ens.gate(x, 0, 0.0)
        torch.cuda.synchronize()
        times.append(np.asarray(ens.profile_read())[-1] * 1e3)
        ens.profile(0)
    t = np.asarray(times[5:])
    print(f"{'dirty' if dirty else 'read-only'} flush, {mode:16s}: median {np.median(t):6.1f} us  min {t.min():6.1f}", flush=True)
