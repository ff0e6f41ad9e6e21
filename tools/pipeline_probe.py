#!/usr/bin/env python3
"""What would two passes in flight be worth?  (diagnostic)  Two independent index handles over the same
synthetic rows and two prober ensembles, passes alternating between two streams, against the same passes
on one stream.  python tools/pipeline_probe.py [rows] [gate_rows] [queries]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import random_prober_state, synth_rows

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_625_000
BG = int(sys.argv[2]) if len(sys.argv) > 2 else 512
BQ = int(sys.argv[3]) if len(sys.argv) > 3 else 64
L, D, DE, K = 6, 2048, 768, 10
ixs, enss, outs = [], [], []
for i in range(2):
    ix = pra.HipFlatIndex(DE, "cos", "f16", capacity=N)
    ix.set_shadow(2)
    ix.add_synthetic(42, 0, N)
    ixs.append(ix)
    e = pra.HipProberEnsemble(L, D, 2, weights="f16")
    for l in range(L):
        e.load_layer(l, random_prober_state(100 + l, D))
    enss.append(e)
    outs.append((torch.empty((L, BG, 2), device="cuda"), torch.empty((BG, 2), device="cuda"),
                 torch.empty((BG,), dtype=torch.int32, device="cuda")))
x = torch.randn((L, BG, D), device="cuda").half()
q = torch.from_numpy(synth_rows(7, 0, BQ, DE)).cuda()
res = [(torch.empty((BQ, K), device="cuda"), torch.empty((BQ, K), dtype=torch.int64, device="cuda")) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(n_pass, two_streams, two_handles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in range(n_pass):
        h = p & 1 if two_handles else 0
        s = streams[p & 1] if two_streams else streams[0]
        with torch.cuda.stream(s):
            enss[h].gate(x, 0, 0.0, out=outs[h])
            ixs[h].search(q, K, out=res[h])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_pass * 1e3


for _ in range(2):
    run(50, False, False); run(50, True, True)
for name, a, b in (("one stream, one handle       ", False, False), ("one stream, handles alternate", False, True),
                   ("two streams, two handles    ", True, True)):
    ms = sorted(run(400, a, b) for _ in range(5))
    print(f"N={N} gate rows={BG} queries={BQ}: {name}: {ms[0]:.4f} .. {ms[-1]:.4f} ms per pass (median {ms[2]:.4f})", flush=True)
D0, I0 = ixs[0].search(q, K)
print("results of the two handles identical:", bool(torch.equal(res[1][1], I0)) and bool(torch.equal(res[0][1], I0)))
