#!/usr/bin/env python3
"""Which (storage, metric, rows, queries) shapes send queries to the exact fallback with the two-level search forced
on (diagnostic; i.i.d. rows: the expected output is just `done`)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import probing_rag_amd as pra
g = torch.Generator(device="cuda").manual_seed(5)
for store in ("f32", "f16"):
  for metric in ("l2", "cos"):
    for N in (33, 500, 4095, 4130, 20000, 65537, 300000):
        for B in (1, 32, 64, 128):
            d, k = 768, 10
            X = torch.randn((N, d), generator=g, device="cuda")
            Q = torch.randn((B, d), generator=g, device="cuda")
            ix = pra.HipFlatIndex(d, metric, store); ix.set_shadow(2); ix.add(X)
            ix.search(Q, k); fb = ix.last_exact_fallbacks()
            if fb: print(store, metric, "N", N, "B", B, "fallbacks", fb, flush=True)
            ix.close()
print("done")
