#!/usr/bin/env python3
"""scan8 kernel time on an 8-GPU-sized shard under the PRAG_SHADOW_DBG timing knobs (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_625_000
ix = pra.HipFlatIndex(768, "cos", "f16", capacity=N)
ix.add_synthetic(42, 0, N)
ix.set_shadow(1)
for B in (64, 32):
    q = torch.from_numpy(synth_rows(7, 0, B, 768)).cuda()
    for _ in range(5): ix.search(q, 10)
    torch.cuda.synchronize()
    ix.profile(256)
    import time
    t0 = time.perf_counter()
    for _ in range(50): ix.search(q, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    k = np.array(ix.profile_read()); ix.profile(0)
    print(f"dbg={os.environ.get('PRAG_SHADOW_DBG','0')} N={N} B={B}: scan8 median {np.median(k)*1e3:.1f} us, search {dt*1e6:.1f} us, ideal {N*776/6.0e12*1e6:.1f} us", flush=True)
