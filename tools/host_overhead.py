#!/usr/bin/env python3
"""Host-side cost of one bench pass (enqueue only) vs its GPU time (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import probing_rag_amd as pra
from probing_rag_amd.synth import random_prober_state, synth_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_625_000
ix = pra.ShardedFlatIndex(768, "cos", "f16", capacity=N)
ix.add_synthetic_local(42, 0, N); ix.sync()
ens = pra.HipProberEnsemble(6, 2048, 2, weights="f16")
for l in range(6): ens.load_layer(l, random_prober_state(100 + l, 2048))
x = torch.randn(6, 512, 2048, device="cuda").half()
q = torch.from_numpy(synth_rows(7, 0, 64, 768)).cuda()
out = (torch.empty(6, 512, 2, device="cuda"), torch.empty(512, 2, device="cuda"), torch.empty(512, dtype=torch.int32, device="cuda"))
def one():
    ens.gate(x, 0, 0.0, out=out)
    return ix.search(q, 10)
for _ in range(50): one()
torch.cuda.synchronize()
n = 500
t0 = time.perf_counter()
for _ in range(n): one()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"N={N}: host enqueue {t_enq / n * 1e6:.1f} us per pass, GPU-complete {t_all / n * 1e6:.1f} us per pass")
# search only
t0 = time.perf_counter()
for _ in range(n): ix.search(q, 10)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"search only: host enqueue {t_enq / n * 1e6:.1f} us, complete {(time.perf_counter() - t0) / n * 1e6:.1f} us")
