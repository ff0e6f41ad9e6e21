#!/usr/bin/env python3
"""The gate at the reference's call shape (B = 1, fp32 weights) 500 times: host-timed, and the program to put after
`rocprofv3 --kernel-trace --stats --` (round 4, one box: 16.6 us per call; small_fc_kernel 8.4 + 6.2 us and
small_head_kernel 4.0 us under the profiler)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import random_prober_state
L, d = 6, 2048
ens = pra.HipProberEnsemble(L, d, 2, weights="f32")
for l in range(L):
    ens.load_layer(l, random_prober_state(100 + l, d))
x = torch.randn((L, 1, d), device="cuda")
out = (torch.empty((L, 1, 2), device="cuda"), torch.empty((1, 2), device="cuda"), torch.empty((1,), dtype=torch.int32, device="cuda"))
for _ in range(50): ens.gate(x, 0, 0.0, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): ens.gate(x, 0, 0.0, out=out)
torch.cuda.synchronize(); print(f"gate B=1 f32: {(time.perf_counter() - t0) / 500 * 1e6:.2f} us per call")
