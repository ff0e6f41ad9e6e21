#!/usr/bin/env python3
"""The INTERLEAVED-cluster corpus of bench.py (row i -> a random centre; CLUSTER_PERIODIC=1: centre i mod 4096, a layout that
resonates with a 256-workgroup grid - all rows of a centre land in two workgroups and overflow their candidate regions; 4 M x 768 fp16 rows, 64 queries near centres):
a few dozen two-level searches - the program to put after `rocprofv3 --kernel-trace --stats --`; prints ms per search,
survivors and the plan.  CLUSTER_ADAPTIVE=0|1, CLUSTER_B=64, CLUSTER_SHADOW=2|0."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import probing_rag_amd as pra
n_rows, n_centres, sigma, d, k = 4_194_304, 4096, 0.0175, 768, 10
B = int(os.environ.get("CLUSTER_B", "64"))
g = torch.Generator(device="cuda").manual_seed(11)
centres = torch.nn.functional.normalize(torch.randn((n_centres, d), generator=g, device="cuda"), dim=1)
ix = pra.HipFlatIndex(d, "cos", "f16", capacity=n_rows)
for lo in range(0, n_rows, 1 << 20):
    m = min(1 << 20, n_rows - lo)
    idx = (torch.arange(lo, lo + m, device="cuda") % n_centres) if os.environ.get("CLUSTER_PERIODIC") == "1" else \
        torch.randint(0, n_centres, (m,), generator=g, device="cuda")
    ix.add(centres[idx] + sigma * torch.randn((m, d), generator=g, device="cuda"))
q = centres[torch.randint(0, n_centres, (B,), generator=g, device="cuda")] + 0.5 * sigma * torch.randn((B, d), generator=g, device="cuda")
ix.set_shadow(int(os.environ.get("CLUSTER_SHADOW", "2")))
ix.set_adaptive(os.environ.get("CLUSTER_ADAPTIVE", "1") == "1")
ix.prepare()
for _ in range(12):
    ix.search(q, k)
    torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for _ in range(n):
    ix.search(q, k)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"B={B} adaptive={os.environ.get('CLUSTER_ADAPTIVE', '1')}: {ms:.3f} ms per search; fallbacks {ix.last_exact_fallbacks()}; "
      f"survivors {ix.last_survivors()}; plan {ix.last_plan()}", flush=True)
