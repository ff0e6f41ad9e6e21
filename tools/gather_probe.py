#!/usr/bin/env python3
"""Where the time of shadow_gather_kernel goes at the 8-GPU shard size: wall time of a 64-query search with
parts of the gather switched off (PRAG_SHADOW_DBG bits 32 = no exact scoring, 64 = no final sort,
128 = no candidate staging; timing only, results wrong).  One process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import time, torch
    import probing_rag_amd as pra
    from probing_rag_amd.synth import synth_rows
    N = int(os.environ.get("PRAG_DOCS", 2_625_000))
    ix = pra.HipFlatIndex(768, "cos", "f16", capacity=N)
    ix.add_synthetic(42, 0, N)
    q = torch.from_numpy(synth_rows(7, 0, 64, 768)).cuda()
    for _ in range(10):
        ix.search(q, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ix.search(q, 10)
    torch.cuda.synchronize()
    print(f"PRAG_SHADOW_DBG={os.environ.get('PRAG_SHADOW_DBG', '0'):>4s}: {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us per search", flush=True)
else:
    for dbg in (0, 32, 64, 96, 224):
        env = dict(os.environ, PRAG_SHADOW_DBG=str(dbg))
        subprocess.run([sys.executable, __file__, "child"], env=env)
