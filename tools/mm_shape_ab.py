#!/usr/bin/env python3
"""MFMA shape of the tiled scans, A/B in ONE process (cdna_hip_programming.md section 5.4 rules 24, 28): the same
searches on two handles of the same corpus, one created under PRAG_MM_SHAPE=32 (v_mfma_f32_32x32x16_f16 /
i32_32x32x32_i8), one under the default 16 x 16 tiles, interleaved rounds, identical ids required.
  python tools/mm_shape_ab.py [rows ...]      (default: 1 000 000 and 21 000 000)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows

sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 21_000_000]
d, k, B = 768, 10, 1000
q = torch.from_numpy(synth_rows(7, 0, B, d)).cuda()


def make(shape, N):
    if shape == 32:
        os.environ["PRAG_MM_SHAPE"] = "32"
    else:
        os.environ.pop("PRAG_MM_SHAPE", None)
    ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
    os.environ.pop("PRAG_MM_SHAPE", None)
    ix.add_synthetic(42, 0, N)
    return ix


def timed(ix, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = ix.search(q, k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


for N in sizes:
    for shadow in ((0,) if N < (2 << 20) else (0, 1)):
        hs = {32: make(32, N), 16: make(16, N)}
        for ix in hs.values():
            ix.set_shadow(2 if shadow else 0)
            ix.prepare()
            for _ in range(3):
                ix.search(q, k)
        reps = 40 if N <= 2_000_000 else 6
        ms = {32: [], 16: []}
        outs = {}
        for rnd in range(7):
            for shape in (32, 16) if rnd % 2 == 0 else (16, 32):
                t, outs[shape] = timed(hs[shape], reps)
                ms[shape].append(t)
        same = bool(torch.equal(outs[32][1], outs[16][1]) and torch.equal(outs[32][0], outs[16][0]))
        m32, m16 = float(np.median(ms[32])), float(np.median(ms[16]))
        print(f"N={N:9d} {'int8 tiles over the shadow' if shadow else 'fp16 tiles':26s}: 32x32 {m32:8.3f} ms (min {min(ms[32]):8.3f}) | "
              f"16x16 {m16:8.3f} ms (min {min(ms[16]):8.3f}) | 16x16 / 32x32 = {m16 / m32:.3f} | identical results: {same} | "
              f"plan {hs[16].last_plan()['family']}", flush=True)
        for ix in hs.values():
            ix.close()
