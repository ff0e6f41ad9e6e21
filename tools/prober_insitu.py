#!/usr/bin/env python3
"""Phase stamps of prober_fused_kernel when it follows a 21 M-row search on the same stream (as in bench.py)
against the same kernel launched back to back.  Needs the diag build:
  PRAG_LIB=probing-rag_amd/lib/libprag_diag.so PRAG_PROBER_STAMPS=1 python tools/prober_insitu.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd import _lib
from probing_rag_amd.synth import random_prober_state, synth_rows

L, B, D = 6, 4096, 2048
N = int(os.environ.get("PRAG_DOCS", 21_000_000))
ens = pra.HipProberEnsemble(L, D, 2, weights="f16")
for l in range(L):
    ens.load_layer(l, random_prober_state(100 + l, D))
x = torch.randn(L, B, D, device="cuda").half()
ix = pra.HipFlatIndex(768, "cos", "f16", capacity=N)
ix.add_synthetic(42, 0, N)
q = torch.from_numpy(synth_rows(7, 0, 64, 768)).cuda()
fn = _lib.lib().prag_diag_prober_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["prologue", "fc1 loop", "stats+sync", "epilogue 1 (cols 0,1)", "publish 0", "sync + LN1 stats",
         "pass 0", "between passes", "pass 1", "epilogue 2 (pass 1)", "sync + logits"]


def stamps():
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (3 * 8 * 32))()
    assert fn(buf, len(buf)) == 0
    return np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(3, 8, 32)


for mode in ("back to back", "after a search"):
    for _ in range(6):
        if mode == "after a search":
            ix.search(q, 10)
        ens.gate(x, 0, 0.0)
    s = stamps()[1]
    t0 = s[:, 0].min()
    print(f"{mode}: total {(s[:, len(names)].max() - t0) / 100:.1f} x100 cycles")
    for i, nm in enumerate(names):
        d = (s[:, i + 1] - s[:, i]) / 100
        print(f"   {nm:24s} {d.mean():7.1f}  (waves {d.min():6.1f} .. {d.max():6.1f})")
