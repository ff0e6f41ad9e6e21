#!/usr/bin/env python3
"""Exact float64 fallback, ms per search for 1 ... 64 flagged queries: one query per pass (exact_scan_kernel), eight per pass
(exact_group_kernel), and the float64 matrix pipe (exact_mfma_kernel: 1-8 flagged queries on four 4 x 4 x 4 blocks, 9-16 on
the 16 x 16 x 4 tile; rows of 1024 elements in groups of 8).  Every query is flagged by construction: 4 M x 640 rows with
k = 30 go straight to the exact scan (no tiled path at that row length); on 2.5 M x 1024 rows every query's row occurs 300
times (more than the tiled first pass's 256 candidates - that pass, ~1.2 ms, is inside the time printed)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import probing_rag_amd as pra
    N, d, k = int(sys.argv[2]), int(sys.argv[3]), 30
    for metric, store in (("l2", "f16"), ("ip", "f16"), ("l2", "f32")):
        ix = pra.HipFlatIndex(d, metric, store, capacity=N + 64 * 300)
        ix.add_synthetic(42, 0, N)
        dup = None
        if d != 640:      # (rows this long have a tiled path at k = 30: flag every query by 300 copies of it among the rows instead)
            dup = torch.randn(64, d, device="cuda")
            ix.add(dup.repeat_interleave(300, dim=0))
        out = []
        for B in (1, 2, 4, 8, 16, 64):
            q = torch.randn(B, d, device="cuda") if dup is None else dup[:B].contiguous()
            for _ in range(2):
                ix.search(q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                ix.search(q, k)
            torch.cuda.synchronize()
            out.append(f"B={B}: {(time.perf_counter() - t0) / n * 1e3:8.2f} ms" + ("" if dup is None else f" ({ix.last_exact_fallbacks()} flagged)"))
        print(f"{metric} {store} {N} x {d}: " + " | ".join(out), flush=True)
        ix.close()
    sys.exit(0)
for N, d in ((4_000_000, 640), (2_500_000, 1024)):      # (the same bytes; rows of 1024: groups of 8 on the matrix pipe)
    for mode, mfma, name in (("0", "0", "one query per pass (exact_scan_kernel)"), ("1", "0", "eight queries per pass (exact_group_kernel)"),
                             ("1", "1", "sixteen (d <= 768) or eight queries per pass on the float64 matrix pipe (exact_mfma_kernel)")):
        if d != 640 and mode == "0":
            continue
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(N), str(d)],
                           env=dict(os.environ, PRAG_EXACT_GROUP=mode, PRAG_EXACT_MFMA=mfma), capture_output=True, text=True)
        print(f"== {name}\n{r.stdout.strip() or r.stderr.strip()[-400:]}", flush=True)
