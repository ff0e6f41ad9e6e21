"""Two-level search at 1 / 32 / 64 / 128 / 1000 queries over N x 768 fp16 rows: ms per search and the scan launch's own
time, for same-box A/B runs of two builds (PRAG_LIB selects the library).  usage: scan8_ab.py [rows] [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import probing_rag_amd as pra  # noqa: E402
from probing_rag_amd.synth import synth_rows  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    d, k = 768, 10
    ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
    ix.add_synthetic(42, 0, N)
    ix.set_shadow(2)
    ix.prepare()
    qs = torch.from_numpy(synth_rows(7, 0, 1000, d)).cuda()
    out = []
    for B in [int(x) for x in os.environ.get("SCAN8_AB_Q", "1,32,64,128,1000").split(",")]:
        q = qs[:B].contiguous()
        for _ in range(3):
            ix.search(q, k)
        torch.cuda.synchronize()
        n = reps if B <= 128 else max(4, reps // 5)
        ix.profile(256)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ix.search(q, k)
        e1.record()
        torch.cuda.synchronize()
        kern = ix.profile_read()
        ix.profile(0)
        ms = e0.elapsed_time(e1) / n
        out.append(f"q{B}={ms:.4f}({np.sum(kern) / n:.4f})")
    print(os.path.basename(os.environ.get("PRAG_LIB", "libprag.so")), N, " ".join(out), "fallbacks", ix.last_exact_fallbacks(), flush=True)


if __name__ == "__main__":
    main()
