#!/usr/bin/env python3
"""Latency of the reference's real call shapes (B=1) on the GPU (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
from tests.golden import cases


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for weights in ("f32", "f16"):
    ens = pra.HipProberEnsemble(6, 2048, 2, weights=weights)
    for l in range(6):
        ens.load_layer(l, cases.synth_state(100 + l, 2048))
    for B in (1, 8, 32, 128, 512, 4096):
        x32 = torch.randn(6, B, 2048, device="cuda")
        x16 = x32.half()
        ens.profile(512)
        t_gate32 = timeit(lambda: ens.gate(x32, 0, 0.0), n=100)
        k32 = np.mean(ens.profile_read())
        t_gate16 = timeit(lambda: ens.gate(x16, 0, 0.0), n=100)
        k16 = np.mean(ens.profile_read())
        ens.profile(0)
        msg = f"weights={weights} B={B:5d}: gate(x f32) {t_gate32:7.1f} us (kernel {k32*1e3:6.1f}) | gate(x f16) {t_gate16:7.1f} us (kernel {k16*1e3:6.1f})"
        if B == 1:
            # the reference's own sequence: six prober(x) calls, each logits.to('cpu') (utils.py:389-390)
            t_ref = timeit(lambda: [p(x32[l]).to("cpu") for l, p in enumerate(ens.probers)], n=50)
            msg += f" | 6 x prober(x).to(cpu) {t_ref:7.1f} us"
        print(msg, flush=True)

for N in (10_000, 1_000_000, 21_000_000):
    ix = pra.HipFlatIndex(768, "l2", "f16", capacity=N)
    ix.add_synthetic(42, 0, N)
    for B in (1, 32):
        q = synth_rows(7, 0, B, 768)
        qd = torch.from_numpy(q).cuda()
        t_dev = timeit(lambda: ix.search(qd, 5), n=30, warm=3)
        t_host = timeit(lambda: ix.search(q, 5), n=30, warm=3)
        print(f"search N={N:9d} B={B:3d} k=5: device-io {t_dev:8.1f} us | numpy-io {t_host:8.1f} us", flush=True)
    del ix
