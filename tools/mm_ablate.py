#!/usr/bin/env python3
"""Where the time of the MFMA-tiled scan goes: C3 (1k x 1M x 768) under timing-only ablations of
scan_mm_kernel (`make -C probing-rag_amd/csrc diag` build; results of ablated runs are wrong)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N, d, B, k = int(os.environ.get("MM_N", 1000000)), 768, int(os.environ.get("MM_B", 1000)), 10
ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
ix.add_synthetic(42, 0, N)
Q = torch.from_numpy(synth_rows(7, 0, B, d)).cuda()
ix.profile(64)
for _ in range(3): ix.search(Q, k)
torch.cuda.synchronize(); ix.profile_read()
t0 = time.perf_counter()
for _ in range(10): ix.search(Q, k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
ker = np.mean(ix.profile_read())
print(f"{os.environ.get('PRAG_MM_ABLATE','0')}: search {dt*1e3:.3f} ms, last-segment kernel {ker:.3f} ms", flush=True)
''' % ROOT
names = {0: "full kernel", 8: "no filter", 9: "no filter, no MFMAs", 10: "no filter, no LDS-DMA",
         12: "no filter, no fragment reads", 14: "MFMAs + barriers only", 15: "barriers only",
         24: "no filter, vmcnt(14)", 40: "no filter, one barrier per phase", 56: "no filter, vmcnt(14), one barrier",
         72: "no filter, int8 MFMA, 768 B rows"}
for abl in (0, 8, 9, 10, 12, 72):
    env = dict(os.environ, PRAG_LIB=os.path.join(ROOT, "probing-rag_amd", "lib", "libprag_diag.so"),
               PRAG_MM_ABLATE=str(abl), PRAG_MM_CLOCK="1")
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    clk = [l for l in out.stderr.splitlines() if "mm diag" in l]
    print(f"{names[abl]:34s}", out.stdout.strip() or out.stderr[-400:], "|", clk[-1] if clk else "", flush=True)
