#!/usr/bin/env python3
"""Phase breakdown of prober_fused_kernel from in-kernel s_memtime stamps (timing-only `make diag` build).

  make -C probing-rag_amd/csrc diag
  PRAG_LIB=probing-rag_amd/lib/libprag_diag.so PRAG_PROBER_STAMPS=1 python tools/prober_stamps.py [d_model]

The hot loops carry no stamps (tools/micro/fc1_loop.hip takes the fc1 loop apart instead).  s_memtime ticks
once per shader cycle; phases print in units of 100 cycles."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd import _lib
from probing_rag_amd.synth import random_prober_state

L, B = 6, 4096
D = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ens = pra.HipProberEnsemble(L, D, 2, weights="f16")
for l in range(L):
    ens.load_layer(l, random_prober_state(100 + l, D))
x = torch.randn(L, B, D, device="cuda").half()
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda") if os.environ.get("PRAG_FLUSH") else None
for _ in range(20):
    if flush is not None:
        flush.add_(1)   # PRAG_FLUSH=1: 2 GiB of traffic between launches (cold L2 / Infinity Cache)
    ens.gate(x, 0, 0.0)
torch.cuda.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * (2 * 3 * 8 * 32))()
# the fp16 x fp16 mode runs prober16.hip (16 x 16 MFMA tiles) unless PRAG_PROBER_SHAPE=32
fn = lib.prag_diag_prober_stamps if os.environ.get("PRAG_PROBER_SHAPE") == "32" else lib.prag_diag_prober16_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, len(buf)) == 0
both = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(2, 3, 8, 32)
s, rt = both[0], both[1]   # s_memtime (core cycles) and s_memrealtime (100 MHz) at the same points
names = ["prologue", "fc1 loop", "stats+sync", "epilogue 1 (cols 0,1)", "publish 0", "sync + LN1 stats",
         "pass 0: k loop | epilogue 1 (cols 2,3)", "sync, publish 1, sync, stats, sync",
         "pass 1: k loop | epilogue 2 (pass 0)", "epilogue 2 (pass 1)", "sync + logits"]
LAST = len(names)
tick = 0.01   # s_memtime ticks once per shader cycle: phases are printed in units of 100 cycles
for sel in range(3):
    t0 = s[sel, :, 0].min()
    dc, dr = s[sel, :, LAST] - s[sel, :, 0], rt[sel, :, LAST] - rt[sel, :, 0]
    print(f"workgroup {sel}: total {(s[sel, :, LAST].max() - t0) * tick:7.2f} x100 cycles; in-kernel clock "
          f"{np.median(dc / np.maximum(dr, 1)) * 100:.0f} MHz over the workgroup's lifetime "
          f"(fc1 loop alone: {np.median((s[sel, :, 2] - s[sel, :, 1]) / np.maximum(rt[sel, :, 2] - rt[sel, :, 1], 1)) * 100:.0f} MHz)")
    for i, nm in enumerate(names):
        d = (s[sel, :, i + 1] - s[sel, :, i]) * tick
        print(f"   {nm:40s} {d.mean():7.2f} c  (waves {d.min():6.2f} .. {d.max():6.2f})   ends at {((s[sel, :, i + 1].max()) - t0) * tick:7.2f}")
if os.environ.get("PRAG_STAMPS_PER_WAVE"):
    for i, nm in enumerate(names):
        print(f"   {nm:40s} per wave:", " ".join(f"{v * tick:7.1f}" for v in (s[1, :, i + 1] - s[1, :, i])))
