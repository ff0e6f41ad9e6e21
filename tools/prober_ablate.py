#!/usr/bin/env python3
"""Upper bound of the epilogue-overlap lever in prober16_kernel: the kernel with its epilogues
stubbed out (loads, LDS staging and every MFMA kept) against the shipped one, B = 4096 and the 512-row slice.
Timing only - the ablated variants compute garbage.  Needs the diag build:
    PRAG_LIB=probing-rag_amd/lib/libprag_diag.so python tools/prober_ablate.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import probing_rag_amd as pra
    from probing_rag_amd.synth import random_prober_state
    L, D = 6, 2048
    ens = pra.HipProberEnsemble(L, D, 2, weights="f16")
    for l in range(L):
        ens.load_layer(l, random_prober_state(100 + l, D))
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    out = []
    for B in (4096, 512):
        x = torch.randn(L, B, D, device="cuda").half()
        for mode in ("b2b", "flushed"):
            ens.profile(256)
            for _ in range(80):
                if mode == "flushed":
                    flush.add_(1)
                ens.gate(x, 0, 0.0)
            torch.cuda.synchronize()
            t = np.asarray(ens.profile_read()) * 1e3
            ens.profile(0)
            out.append(f"B={B} {mode} {np.median(t):6.1f} us")
    print(" | ".join(out), flush=True)
    sys.exit(0)
names = {0: "shipped", 1: "no epilogue 1", 2: "no publish", 4: "no epilogue 2", 8: "no LN statistics", 5: "no epilogue 1+2",
         7: "no epilogue 1+2, no publish", 15: "MFMAs + loads + staging only"}
for mask, name in names.items():
    env = dict(os.environ, PRAG_PROBER_ABLATE=str(mask))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    print(f"ablate {mask:2d} ({name:32s}): {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
