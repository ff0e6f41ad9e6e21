#!/usr/bin/env python3
"""Differential fuzz of the large-batch / deep-list paths (diagnostic): the same rows in two indexes, one taking the
MFMA-tiled scan above 128 queries, the other created under PRAG_SCAN_MM=0 (per-lane-list / query-stationary
kernels), a third one with the 8-bit shadow forced on (shadow mode 2: above 128 queries the int8 tiles select first,
the fp16 tiles repeat a batch in which a query failed the 8-bit certificate); every path is exact, so ids and scores
must be identical.  python tools/fuzz_paths.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n_case, n_fb = time.time(), 0, 0
n_t8, n_t8_second = 0, 0
by = {}
while time.time() - t0 < budget:
    d = int(rng.choice([256, 512, 640, 768, 1024]))
    N = int(rng.choice([300, 2048, 2049, 5000, 40_000, 150_000, 400_000], p=[.1, .1, .1, .2, .25, .15, .1]))
    B = int(rng.choice([129, 130, 200, 256, 257, 400, 700]))
    k = int(rng.choice([1, 5, 10, 26, 27, 40, 100], p=[.1, .2, .3, .1, .1, .1, .1]))
    metric = str(rng.choice(["l2", "ip", "cos"]))
    store = str(rng.choice(["f16", "f32"]))
    kind = str(rng.choice(["iid", "clustered", "dups", "sorted"]))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    X = torch.randn((N, d), generator=g, device="cuda")
    Q = torch.randn((B, d), generator=g, device="cuda")
    if kind == "clustered":
        nc = max(1, N // 257)
        c = torch.randn((nc, d), generator=g, device="cuda")
        X = c[(torch.arange(N, device="cuda") * nc) // N] + 0.05 * X
    elif kind == "dups":
        X[N // 2:] = X[: N - N // 2].clone()
    elif kind == "sorted":          # later rows match query 0 better: overflows threshold-filtered candidate stores
        s_ = X @ Q[0]
        X = X[torch.argsort(s_)].contiguous()
    Q[1] = X[N // 3]
    os.environ.pop("PRAG_SCAN_MM", None)
    a = pra.HipFlatIndex(d, metric, store)
    a.add(X)
    os.environ["PRAG_SCAN_MM"] = "0"
    b = pra.HipFlatIndex(d, metric, store)
    b.add(X)
    os.environ.pop("PRAG_SCAN_MM", None)
    Da, Ia = a.search(Q, k)
    fb = a.last_exact_fallbacks()
    n_fb += fb
    key = (kind, metric, store, 'exact-only' if (k > 26 and d == 640) else 'k>26' if k > 26 else 'k<=26', 'N<=5k' if N <= 5000 else 'N>5k')
    c_ = by.setdefault(key, [0, 0, 0]); c_[0] += 1; c_[1] += B; c_[2] += fb
    Db, Ib = b.search(Q, k)
    n_case += 1
    c8 = pra.HipFlatIndex(d, metric, store)
    c8.set_shadow(2)
    c8.add(X)
    Dc, Ic = c8.search(Q, k)
    t8 = c8.last_tiled8()
    n_t8 += t8 >= 0
    n_t8_second += t8 > 0
    if not (torch.equal(Ic, Ib) and torch.equal(Dc, Db)):
        bad = (Ic != Ib).nonzero()[:5].tolist()
        print(f"MISMATCH (int8 tiles, failed first tier: {t8}) d={d} N={N} B={B} k={k} {metric} {store} {kind}: id diffs {bad}, "
              f"max |dD| {float((Dc - Db).abs().max()):.3e}", flush=True)
        sys.exit(1)
    c8.close()
    if not (torch.equal(Ia, Ib) and torch.equal(Da, Db)):
        bad = (Ia != Ib).nonzero()[:5].tolist()
        print(f"MISMATCH d={d} N={N} B={B} k={k} {metric} {store} {kind}: id diffs {bad}, max |dD| {float((Da - Db).abs().max()):.3e}", flush=True)
        for b_, j_ in bad[:3]:
            print("   ", b_, j_, "tiled", int(Ia[b_, j_]), float(Da[b_, j_]), "lists", int(Ib[b_, j_]), float(Db[b_, j_]), flush=True)
        sys.exit(1)
    a.close(); b.close()
for key in sorted(by, key=lambda k_: -by[k_][2] / max(1, by[k_][1]))[:22]:
    c_ = by[key]
    print(key, f"cases {c_[0]} queries {c_[1]} fallbacks {c_[2]} = {c_[2] / max(1, c_[1]):.3f} per query")
print(f"fuzz ok: {n_case} cases in {time.time() - t0:.0f} s, {n_fb} exact fallbacks on the tiled path; "
      f"{n_t8} cases took the int8 tiles, {n_t8_second} of them needed the second tier", flush=True)
