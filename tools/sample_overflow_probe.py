#!/usr/bin/env python3
"""How often does a 64-query two-level search send a query to the exact scan (candidate-region overflow) with the sampled
pre-bound on / off?  2.625 M x 768 fp16 rows, fresh random queries every search.  PRAG_SHADOW_SAMPLE=0|1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N = int(os.environ.get("PRAG_DOCS", 2_625_000))
ix = pra.HipFlatIndex(768, "cos", "f16", capacity=N)
ix.add_synthetic(42, 0, N)
ix.set_shadow(1); ix.prepare()
n, fb, t = int(os.environ.get("N_SEARCH", 300)), 0, 0.0
for i in range(n):
    q = torch.from_numpy(synth_rows(1000 + i, 0, 64, 768)).cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix.search(q, 10)
    torch.cuda.synchronize(); t += time.perf_counter() - t0
    fb += ix.last_exact_fallbacks()
print(f"PRAG_SHADOW_SAMPLE={os.environ.get('PRAG_SHADOW_SAMPLE', 'default')}: {n} searches x 64 queries over {N} rows: "
      f"{fb} queries took the exact scan; {t / n * 1e3:.3f} ms per search (host-timed, synchronous)", flush=True)
