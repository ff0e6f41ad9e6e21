#!/usr/bin/env python3
"""Differential fuzz of the two-level search against the direct scan, both on the GPU (diagnostic):
random (rows, queries, k, d, metric, storage, corpus structure); ids must be identical, scores equal to 1e-6
relative.  python tools/fuzz_shadow.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra

from tests.golden import cases as _cases

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n_case, n_fb = time.time(), 0, 0
by = {}
# prag_search_and_gate sub-cases: one prober ensemble, the gate's launch beside the bound kernel (PRAG_SCAN_GATE=0), behind
# the scan's workgroups (1, whenever the shape has that form) or where the library puts it (-1)
_pc = _cases.PROBER_CASES[1]
_ens = pra.HipProberEnsemble(_pc["L"], _pc["d"], 2, weights="f16")
for _l in range(_pc["L"]):
    _ens.load_layer(_l, _cases.synth_state(_pc["wseed"] + _l, _pc["d"]))
_gate_x = {}
n_fused = 0
while time.time() - t0 < budget:
    os.environ["PRAG_SCAN_GATE"] = str(rng.choice(["0", "1", "-1"]))     # (read when an index is created)
    os.environ["PRAG_SCAN_WG_TUNE"] = str(rng.choice(["0", "1", "-1"]))
    d = int(rng.choice([128, 256, 384, 512, 640, 768, 1024]))
    N = int(rng.choice([1, 7, 33, 500, 4095, 4096, 4130, 20_000, 65_537, 300_000, 1_200_000], p=[.03, .03, .04, .1, .1, .1, .1, .2, .15, .1, .05]))
    if N * d > 600_000_000:
        N = 600_000_000 // d
    B = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128], p=[.15, .05, .05, .1, .1, .05, .15, .1, .05, .05, .15]))
    k = int(rng.choice([1, 5, 10, 12, 13, 26]))
    metric = str(rng.choice(["l2", "ip", "cos"]))
    store = str(rng.choice(["f16", "f32"]))
    kind = str(rng.choice(["iid", "clustered", "dups", "scaled", "embedded"]))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    X = torch.randn((N, d), generator=g, device="cuda")
    if kind == "clustered" and N >= 64:
        nc = max(1, N // 257)
        c = torch.randn((nc, d), generator=g, device="cuda")
        X = c[(torch.arange(N, device="cuda") * nc) // N] + 0.02 * X
    elif kind == "dups" and N >= 8:
        X[N // 2:] = X[: N - N // 2].clone()
    elif kind == "scaled":
        X = X * torch.exp(torch.randn((N, 1), generator=g, device="cuda"))
    Q = torch.randn((B, d), generator=g, device="cuda")
    if kind == "embedded":
        # a common mean direction and a few outlier coordinates (the geometry of real sentence embeddings): the shadow's
        # affine map, the per-row bias term and the centred queries carry this case; queries share the structure
        mu = torch.randn((d,), generator=g, device="cuda") * float(rng.choice([0.5, 2.0, 8.0]))
        cols = torch.randperm(d, generator=g, device="cuda")[: int(rng.integers(1, 8))]
        mu[cols] = mu[cols].sign() * float(rng.choice([10.0, 30.0, 100.0]))
        jit = torch.zeros((d,), device="cuda")
        jit[cols] = float(rng.choice([0.0, 0.1, 0.5]))
        X = X + mu * (1.0 + jit * torch.randn((N, 1), generator=g, device="cuda"))
        if rng.random() < 0.7:
            Q = Q + mu * (1.0 + jit * torch.randn((B, 1), generator=g, device="cuda"))
    if N >= 4:
        Q[0] = X[N // 3]
        Q[B - 1] = X[N - 1] + 0.01 * Q[B - 1]
    ix = pra.HipFlatIndex(d, metric, store)
    half = N // 2
    ix.set_shadow(2)
    if half:
        ix.add(X[:half])
    ix.add(X[half:])
    D1, I1 = ix.search(Q, k)
    fb = ix.last_exact_fallbacks()
    ix.set_shadow(0)
    D0, I0 = ix.search(Q, k)
    ok = torch.equal(I0, I1) and torch.allclose(D0, D1, rtol=1e-6, atol=0)
    if ok and N >= 3 and rng.random() < 0.15:
        # the same rows as 2-3 row shards with the two-level search forced on, searched shard by shard with global
        # ids and merged (the multi-GPU data path minus the collective): must equal the unsharded answer
        cuts = sorted(set([0, N] + [int(x) for x in rng.integers(1, N, size=int(rng.integers(1, 3)))]))
        shards = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            sh = pra.HipFlatIndex(d, metric, store)
            sh.set_shadow(2)
            sh.add(X[lo:hi])
            shards.append(sh)
        Ds, Is = pra.search_shards_on_one_gpu(shards, Q, k, metric)
        # (ids cross shards tagged with the float32 residual of their float64 score - prag_index_search_tagged -
        # so even a float32 tie across shards comes out in the unsharded order)
        ok = torch.equal(Ds, D0) and torch.equal(Is, I0)
        for sh in shards:
            sh.close()
        n_sharded = globals().get("n_sharded", 0) + 1
        globals()["n_sharded"] = n_sharded
    if ok and rng.random() < 0.2:
        # the search AND the gate of another batch in one call: D / I those of the two-level search, logits / sums /
        # decisions those of ens.gate, whichever launch carries the prober's workgroups
        Bg = int(rng.choice([40, 96, 512, 1400]))
        if Bg not in _gate_x:
            _gate_x[Bg] = torch.from_numpy(_cases.synth_x(_pc["xseed"] + Bg, _pc["L"], Bg, _pc["d"], 1.0)).cuda().half()
        abl, theta = int(rng.integers(0, 3)), float(rng.choice([0.0, 0.25]))
        want = [t.clone() for t in _ens.gate(_gate_x[Bg], abl, theta)]
        ix.set_shadow(2)
        (D2, I2), got = pra.search_and_gate(ix, Q, k, _ens, _gate_x[Bg], abl, theta)
        ok = torch.equal(I2, I1) and torch.equal(D2, D1) and all(torch.equal(a, b) for a, b in zip(got, want))
        if not ok:
            print(f"MISMATCH in search_and_gate (PRAG_SCAN_GATE={os.environ['PRAG_SCAN_GATE']}, gate rows {Bg}): plan {ix.last_plan()}", flush=True)
        n_fused += 1
    n_case += 1
    if 'Is' in dir() and ok:
        del Is
    n_fb += fb
    key = (kind, metric, store, 'N<5k' if N < 5000 else 'N<100k' if N < 100_000 else 'N>=100k', 'B<=32' if B <= 32 else 'B<=64' if B <= 64 else 'B<=128')
    c = by.setdefault(key, [0, 0, 0])
    c[0] += 1; c[1] += B; c[2] += fb
    if not ok:
        bad = (I0 != I1).nonzero()[:5].tolist()
        print(f"MISMATCH d={d} N={N} B={B} k={k} {metric} {store} {kind}: two-level vs direct id diffs {bad}, "
              f"max |dD| {float((D0 - D1).abs().max()):.3e} (fallbacks {fb})", flush=True)
        if "Is" in dir():
            bs = (Is != I0).nonzero()[:5].tolist()
            print(f"  sharded sub-case: cuts {cuts}, id diffs {bs}, max |dD| {float((Ds - D0).abs().max()):.3e}", flush=True)
            for b_, j_ in bs[:3]:
                print("   ", b_, j_, "whole", int(I0[b_, j_]), float(D0[b_, j_]), "sharded", int(Is[b_, j_]), float(Ds[b_, j_]), flush=True)
        sys.exit(1)
    ix.close()
for key in sorted(by, key=lambda k: -by[k][2] / max(1, by[k][1]))[:25]:
    c = by[key]
    print(key, f"cases {c[0]} queries {c[1]} fallbacks {c[2]} = {c[2] / max(1, c[1]):.3f} per query")
print(f"sharded sub-cases: {globals().get('n_sharded', 0)}; search_and_gate sub-cases: {n_fused}")
print(f"fuzz ok: {n_case} cases in {time.time() - t0:.0f} s, {n_fb} exact fallbacks in total", flush=True)
