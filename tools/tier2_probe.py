#!/usr/bin/env python3
"""Second tier of the int8 tiles when a FEW queries fail their certificate (diagnostic): N synthetic rows + one cluster
of 700 look-alikes, 1000 queries of which 3 sit in the cluster.  python tools/tier2_probe.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N, d, B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000, 768, 1000, 10
ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N + 700)
ix.add_synthetic(42, 0, N)
g = torch.Generator(device="cuda").manual_seed(3)
base = torch.randn((1, d), generator=g, device="cuda")
ix.add(base + 2e-3 * torch.randn((700, d), generator=g, device="cuda"))
Q = torch.from_numpy(synth_rows(7, 0, B, d)).cuda()
Qc = Q.clone()
for i in (5, 500, 999):
    Qc[i] = (base + 2e-3 * torch.randn((1, d), generator=g, device="cuda"))[0]
for name, q in (("no query in the cluster", Q), ("3 queries in the cluster", Qc)):
    for shadow in (0, 1):
        ix.set_shadow(shadow); ix.prepare()
        for _ in range(2):
            D, I = ix.search(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            D, I = ix.search(q, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        if shadow == 0:
            I_ref = I.clone()
        print(f"N={N}+700 {name}: {'int8 tiles first' if shadow else 'fp16 tiles      '}: {ms:.2f} ms per search, failed first tier {ix.last_tiled8()}, "
              f"exact fallbacks {ix.last_exact_fallbacks()}" + ("" if shadow == 0 else f", ids identical {bool(torch.equal(I, I_ref))}"), flush=True)
