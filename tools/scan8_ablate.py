#!/usr/bin/env python3
"""Where the time of scan8_kernel goes (timing-only `make diag` build, results WRONG): PRAG_SHADOW_DBG bits 1024 = no
epilogue arithmetic, 2048 = no fragment reads / MFMAs, 4096 = the stream alone (loads into registers).  One process per
setting; prints the scan kernel's own time (event ring) for 64 queries."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import probing_rag_amd as pra
    from probing_rag_amd.synth import synth_rows
    N = int(os.environ.get("PRAG_DOCS", 21_000_000))
    ix = pra.HipFlatIndex(768, "cos", "f16", capacity=N)
    ix.add_synthetic(42, 0, N)
    ix.set_shadow(1)
    ix.prepare()
    nq = int(os.environ.get("PRAG_QUERIES", 64))
    q = torch.from_numpy(synth_rows(7, 0, nq, 768)).cuda()
    for _ in range(5):
        ix.search(q, 10)
    torch.cuda.synchronize()
    ix.profile(64)
    for _ in range(30):
        ix.search(q, 10)
    torch.cuda.synchronize()
    ms = np.asarray(ix.profile_read())
    import time as _t
    t0 = _t.perf_counter()
    for _ in range(20):
        ix.search(q, 10)
    torch.cuda.synchronize()
    wall = (_t.perf_counter() - t0) / 20 * 1e3
    alg = N * (768 + 8)
    print(f"PRAG_SHADOW_DBG={os.environ.get('PRAG_SHADOW_DBG', '0'):>5s}: scan8 {ms.mean():.4f} ms -> {alg / ms.mean() / 1e9:.2f} TB/s "
          f"({alg / ms.mean() / 1e9 / 8:.3f} of 8); search {wall:.3f} ms", flush=True)
else:
    env0 = dict(os.environ, PRAG_LIB=os.path.join(ROOT, "probing-rag_amd", "lib", "libprag_diag.so"))
    for dbg in [int(x) for x in os.environ.get('ABLATE', '0,1024,2048,3072,4096,5120').split(',')]:
        subprocess.run([sys.executable, __file__, "child"], env=dict(env0, PRAG_SHADOW_DBG=str(dbg)))
