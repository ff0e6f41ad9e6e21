"""GPU: two-level exact search through the 8-bit shadow (prag_index_set_shadow, flat_shadow.hip).
The shadow only decides which rows need not be looked at: results must be the float64 definition's,
bit for bit, for every shape - and the int8 matrix-core operand layout is pinned by them."""
import numpy as np
import pytest

from oracle import oracle_c, oracle_np as onp

pytestmark = pytest.mark.gpu

METRICS = [onp.METRIC_L2, onp.METRIC_IP, onp.METRIC_COS]


@pytest.fixture(params=[None, "1"], ids=["bound-auto", "bound-on"])
def shadow_bound(request, monkeypatch):
    """Second run with the exact-bound / finish kernel in front of the gather (shadow_bound_kernel; by itself from 2^19
    rows) and the quad-test scan kernels (from 2^23 rows) forced on at every size - the big-corpus tests and the bench
    riders cover the default side."""
    if request.param:
        monkeypatch.setenv("PRAG_SHADOW_BOUND", request.param)     # both read when the index is created
        monkeypatch.setenv("PRAG_SCAN8_QUAD_ROWS", "0")
    return request.param


def _stored(X, metric, store):
    xs = onp.normalize_rows(X) if metric == onp.METRIC_COS else X
    return onp.store_round(xs, store)


def _check(D, I, D0, I0, metric):
    assert np.array_equal(I, I0), np.argwhere(I != I0)[:4]
    if metric == onp.METRIC_L2:
        np.testing.assert_allclose(D, D0, rtol=1e-4, atol=1e-6)
    else:
        np.testing.assert_allclose(D, D0, atol=1e-4, rtol=0)


@pytest.mark.parametrize("store", ["f16", "f32"])
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("N,B,k,d", [(10_000, 64, 5, 768), (1, 1, 5, 768), (31, 3, 10, 128), (2049, 33, 10, 256),
                                     (4097, 40, 12, 512), (777, 1, 1, 1024), (30_000, 7, 10, 768),
                                     (20_000, 3, 26, 256), (9000, 50, 20, 768),      # 32-deep bound lists
                                     (5000, 70, 5, 384)])
def test_shadow_search_matches_definition(metric, store, N, B, k, d, shadow_bound):
    import probing_rag_amd as pra
    X = onp.synth_rows(42, 0, N, d)
    if N > 40:
        X[N // 2] = X[3]
        X[N - 1] = X[3]
    Q = onp.synth_rows(7, 0, B, d)
    if N > 40:
        Q[0] = X[3]
    ix = pra.HipFlatIndex(d, metric, store)
    ix.set_shadow(2)
    ix.add(X[: N // 2])
    ix.add(X[N // 2:])                         # the shadow follows incremental adds
    D, I = ix.search(Q, k)
    D0, I0 = oracle_c.flat_search(_stored(X, metric, store), Q, k, metric)
    _check(D, I, D0, I0, metric)
    ix.add(X[:100] * np.float32(1.5))          # grow after a search: rows 0..99 scaled get new ids
    X2 = np.concatenate([X, X[:100] * np.float32(1.5)])
    D, I = ix.search(Q, k)
    D0, I0 = oracle_c.flat_search(_stored(X2, metric, store), Q, k, metric)
    _check(D, I, D0, I0, metric)


@pytest.mark.parametrize("metric", METRICS)
def test_shadow_on_rows_the_quantiser_handles_badly(metric, shadow_bound):
    """Near-parallel rows with one huge element each: the per-row 8-bit grid is coarse relative to the
    differences that decide the ranking, the filter cannot exclude much, candidate regions overflow -
    flagged queries must come out of the exact scan; results still exact."""
    import probing_rag_amd as pra
    N, d, B, k = 40_000, 256, 20, 10
    rng = np.random.default_rng(3)
    base = onp.synth_rows(5, 0, 1, d)[0]
    X = (base[None, :] + 1e-3 * rng.standard_normal((N, d))).astype(np.float32)
    X[np.arange(N), rng.integers(0, d, N)] += 50.0           # one outlier element per row sets the scale
    Q = (base[None, :] + 1e-3 * rng.standard_normal((B, d))).astype(np.float32)
    for store in ("f16", "f32"):
        ix = pra.HipFlatIndex(d, metric, store)
        ix.set_shadow(2)
        ix.add(X)
        D, I = ix.search(Q, k)
        D0, I0 = oracle_c.flat_search(_stored(X, metric, store), Q, k, metric)
        _check(D, I, D0, I0, metric)
        ix.close()


def test_shadow_filter_statistics_on_a_random_corpus():
    """1 M exchangeable rows: the filter must leave a few hundred candidates per query, no overflow
    (no exact fallback), and agree with the plain fp16 scan bit for bit."""
    import torch
    import probing_rag_amd as pra
    N, d, k = 1_200_000, 768, 10       # >= 2^20 rows: mode 1 switches the shadow on
    ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
    ix.add_synthetic(42, 0, N)
    Q = onp.synth_rows(7, 0, 64, d)
    for i in range(8):
        Q[i] = onp.synth_rows(42, 1000 + 7919 * i, 1, d)[0] + 0.05 * Q[i]
    q = torch.from_numpy(Q).cuda()
    for B in (64, 1, 32):
        D0, I0 = ix.search(q[:B], k)
        ix.set_shadow(1)
        D1, I1 = ix.search(q[:B], k)
        assert ix.last_exact_fallbacks() == 0
        ix.set_shadow(0)
        assert torch.equal(I0, I1)
        assert torch.allclose(D0, D1, rtol=1e-6, atol=0)
    assert (I0[:1, 0].cpu().numpy() == [1000]).all()
    ix.set_shadow(1)
    Dn, In = ix.search(Q[:5], 5)                                  # NumPy in/out (host i/o path)
    ix.set_shadow(0)
    Dm, Im = ix.search(Q[:5], 5)
    assert np.array_equal(In, Im) and np.allclose(Dn, Dm, rtol=1e-6)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_shadow_equals_the_direct_scan_on_clustered_rows(seed, shadow_bound):
    """Mid-size shards (a few tiles per wave: the warm-up / second-visit logic and the early bound slots
    carry the whole search) with clustered, unevenly scaled rows: the two-level search and the direct
    scan of the stored rows are independent routes to the same answer and must agree bit for bit; a few
    queries are also checked against the float64 oracle."""
    import torch
    import probing_rag_amd as pra
    rng = np.random.default_rng(100 + seed)
    d, k = 768, 10
    N = int(rng.choice([70_000, 200_000, 400_000]))
    n_clusters = 40
    centres = rng.standard_normal((n_clusters, d)).astype(np.float32) * rng.uniform(0.5, 3.0, (n_clusters, 1)).astype(np.float32)
    which = rng.integers(0, n_clusters, N)
    X = centres[which] + (0.15 * rng.standard_normal((N, d))).astype(np.float32)
    X *= rng.lognormal(0.0, 0.5, (N, 1)).astype(np.float32)            # norm spread
    B = 64
    Q = centres[rng.integers(0, n_clusters, B)] + (0.15 * rng.standard_normal((B, d))).astype(np.float32)
    Q[:8] = X[rng.integers(0, N, 8)] * np.float32(1.001)                # near-copies of stored rows
    for metric, store in ((onp.METRIC_COS, "f16"), (onp.METRIC_L2, "f32"), (onp.METRIC_IP, "f16")):
        ix = pra.HipFlatIndex(d, metric, store, capacity=N)
        ix.add(X)
        qd = torch.from_numpy(Q).cuda()
        out = {}
        for mode in (0, 2):
            ix.set_shadow(mode)
            for Bq in (64, 32, 5):
                Dm, I = ix.search(qd[:Bq], k)
                out[(mode, Bq)] = (Dm.cpu().numpy(), I.cpu().numpy())
        for Bq in (64, 32, 5):
            assert np.array_equal(out[(0, Bq)][1], out[(2, Bq)][1]), (metric, store, Bq)
            np.testing.assert_allclose(out[(0, Bq)][0], out[(2, Bq)][0], rtol=1e-6, atol=1e-6)
        D0, I0 = oracle_c.flat_search(_stored(X, metric, store), Q[:6], k, metric)
        _check(out[(2, 64)][0][:6], out[(2, 64)][1][:6], D0, I0, metric)
        ix.close()


def test_contiguous_clusters_stay_on_the_two_level_path():
    """A corpus in 'article order' - 1024 contiguous rows around each centre, cosine ~0.9 to it - with
    queries next to centres: thousands of rows sit inside the shadow's error band, and a loosely bounded
    query sees whole tiles of look-alikes at once.  Results must equal the direct scan bit for bit, and the
    candidate regions must hold (round 2's 128-slot regions and wave-major tile map sent all 64 queries to
    the exact float64 scan: 104 ms per search against 1.1 ms for the direct scan)."""
    import torch
    import probing_rag_amd as pra
    d, n_rows, n_centres, sigma, B, k = 768, 1 << 20, 1024, 0.0175, 64, 10
    g = torch.Generator(device="cuda").manual_seed(11)
    centres = torch.nn.functional.normalize(torch.randn((n_centres, d), generator=g, device="cuda"), dim=1)
    ix = pra.HipFlatIndex(d, "cos", "f16", capacity=n_rows)
    for lo in range(0, n_rows, 1 << 18):
        idx = (torch.arange(lo, lo + (1 << 18), device="cuda") * n_centres) // n_rows
        ix.add(centres[idx] + sigma * torch.randn((1 << 18, d), generator=g, device="cuda"))
    q = centres[torch.randint(0, n_centres, (B,), generator=g, device="cuda")] + \
        0.5 * sigma * torch.randn((B, d), generator=g, device="cuda")
    ix.set_shadow(0)
    D0, I0 = ix.search(q, k)
    ix.set_shadow(2)
    ix.prepare()
    for nq in (64, 32, 1):
        D1, I1 = ix.search(q[:nq], k)
        assert torch.equal(I1, I0[:nq]) and torch.allclose(D1, D0[:nq], rtol=1e-6, atol=0)
        assert ix.last_exact_fallbacks() <= nq // 16
    ix.close()


@pytest.mark.parametrize("store", ["f16", "f32"])
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("N,B,k,d", [(40_000, 128, 10, 768), (9_001, 65, 5, 768), (20_000, 100, 12, 512),
                                     (300, 77, 10, 768)])
def test_128_query_tiles_match_definition(metric, store, N, B, k, d, shadow_bound):
    """65..128 queries take ONE pass over the shadow with 128-query tiles (list-less scan8, bound from the
    slot epochs alone): results are the float64 definition's bit for bit, duplicates and planted rows included."""
    import probing_rag_amd as pra
    X = onp.synth_rows(42, 0, N, d)
    X[N // 2] = X[3]
    X[N - 1] = X[3]
    Q = onp.synth_rows(7, 0, B, d)
    Q[0] = X[3]
    Q[B - 1] = X[N // 3] + np.float32(0.01) * Q[B - 1]
    ix = pra.HipFlatIndex(d, metric, store)
    ix.set_shadow(2)
    ix.add(X)
    D, I = ix.search(Q, k)
    D0, I0 = oracle_c.flat_search(_stored(X, metric, store), Q, k, metric)
    _check(D, I, D0, I0, metric)
    # the same queries through 64-query tiles (two passes) and directly over the stored rows: identical
    import torch
    qd = torch.from_numpy(Q).cuda()
    Da, Ia = ix.search(qd[:64], k)
    assert np.array_equal(Ia.cpu().numpy(), I[:64])
    ix.set_shadow(0)
    Db, Ib = ix.search(qd, k)
    assert np.array_equal(Ib.cpu().numpy(), I)
    ix.close()


def test_sliced_gather_merge_never_reads_a_stale_list(monkeypatch):
    """The 16 slices of a query publish their lists across XCDs with relaxed device-scope stores and the last one to
    arrive merges them (flat_shadow.hip, end of shadow_gather_kernel).  Two different query batches alternate on the
    same index 150 times, so every merge finds the OTHER batch's lists in part_key / part_id from the search before:
    one stale read gives a wrong id.  The bound kernel is off here (it would finish these queries itself)."""
    import torch
    import probing_rag_amd as pra
    monkeypatch.setenv("PRAG_SHADOW_BOUND", "0")
    N, d, k = 60_000, 768, 10
    ix = pra.HipFlatIndex(d, onp.METRIC_COS, "f16")
    ix.set_shadow(2)
    ix.add_synthetic(42, 0, N)
    QA = torch.from_numpy(onp.synth_rows(7, 0, 64, d)).cuda()
    QB = torch.from_numpy(onp.synth_rows(8, 0, 64, d)).cuda()
    DA, IA = (t.clone() for t in ix.search(QA, k))
    DB, IB = (t.clone() for t in ix.search(QB, k))
    X = ix.reconstruct_n(0, N)
    for Q, I0 in ((QA, IA), (QB, IB)):
        _, I_ref = oracle_c.flat_search(X, Q.cpu().numpy(), k, onp.METRIC_COS)
        assert np.array_equal(I0.cpu().numpy(), I_ref)
    bad = 0
    for it in range(150):
        for Q, D0, I0 in ((QA, DA, IA), (QB, DB, IB)):
            D, I = ix.search(Q, k)
            bad += int(not (torch.equal(I, I0) and torch.equal(D, D0)))
    torch.cuda.synchronize()
    assert bad == 0


@pytest.mark.parametrize("metric,store", [("cos", "f16"), ("l2", "f16"), ("l2", "f32")])
@pytest.mark.parametrize("kw", [{}, {"n_outlier": 4, "outlier_ratio": (10.0, 14.0)}], ids=["6_outliers_10-30x", "4_outliers_dense_mean"])
def test_embedding_shaped_corpus_is_searched_exactly_and_tightly(metric, store, kw):
    """un-normalised sentence-embedding geometry (make_indexer.py:447-456 indexes contriever output under
    L2): rows share a mean at 0.8 of their norm, a power-law spectrum, outlier coordinates at 10-30 x the median.
    1 Mi rows, ids bit-exact against the C oracle on the stored rows for 64 queries and for the reference's single
    query; the two-level search must stay a FILTER on such rows - no exact fallback, survivors per query in the
    hundreds, not the tens of thousands a per-row abs-max grid let through (the shadow quantises (x - mu) / c)."""
    import torch
    import probing_rag_amd as pra
    from probing_rag_amd.synth import embedding_like_rows, embedding_structure
    N, d, k = 1 << 20, 768, 10
    st = embedding_structure(5, d, **kw)
    ix = pra.HipFlatIndex(d, metric, store, capacity=N)
    for lo in range(0, N, 1 << 18):
        ix.add(embedding_like_rows(5, lo, 1 << 18, d, structure=st))
    q = embedding_like_rows(1005, 0, 64, d, structure=st)
    mid = {"l2": onp.METRIC_L2, "cos": onp.METRIC_COS}[metric]
    stored = ix.reconstruct_n(0, N)                         # what the index holds (normalised / rounded)
    D0, I0 = oracle_c.flat_search(stored, q.cpu().numpy(), k, mid)
    ix.set_shadow(2)
    ix.prepare()
    for B in (64, 1):
        D, I = ix.search(q[:B], k)
        assert ix.last_plan()["family"] == "scan8_kernel"
        _check(D.cpu().numpy(), I.cpu().numpy(), D0[:B], I0[:B], mid)
        assert ix.last_exact_fallbacks() == 0
        sv = ix.last_survivors()
        # (regions hold 131 072 per query; before the affine map: 180 000 - 870 000 per query and every query in the exact
        #  scan.  The tail - a query with a 3-sigma jitter on an outlier coordinate - is what the mean leaves over.)
        assert sv["queries"] == B and sv["max_per_query"] < 131_072, sv
        assert sv["per_query"] < 15_000, sv
    # the direct scan of the stored rows agrees too (its certificate has to cope with the common mean)
    ix.set_shadow(0)
    D, I = ix.search(q, k)
    _check(D.cpu().numpy(), I.cpu().numpy(), D0, I0, mid)
    # > 128 queries on the int8 tiles over the same shadow: the certificate compares keys in the shifted key space
    ix.set_shadow(2)
    q300 = embedding_like_rows(1006, 0, 300, d, structure=st)
    D3, I3 = ix.search(q300, k)
    D30, I30 = oracle_c.flat_search(stored, q300.cpu().numpy(), k, mid)
    _check(D3.cpu().numpy(), I3.cpu().numpy(), D30, I30, mid)
    ix.close()


def test_affine_shadow_map_is_a_performance_choice_only(monkeypatch):
    """PRAG_SHADOW_AFFINE=0 keeps the round 2-4 shadow (mu = 0, c = 1), 2 adds column scales: same ids, same scores - on rows with a common
    mean and outlier coordinates it lets far more rows through; rows added AFTER the map was fitted (a different
    distribution, even) are still searched exactly, and so is a shadow rebuilt when the capacity grows."""
    import torch
    import probing_rag_amd as pra
    from probing_rag_amd.synth import embedding_like_rows, embedding_structure
    N, d, k = 200_000, 768, 10
    st = embedding_structure(9, d)
    X = embedding_like_rows(9, 0, N, d, structure=st)
    q = embedding_like_rows(1009, 0, 40, d, structure=st)
    res, surv = {}, {}
    for mode in ("1", "0", "2"):
        monkeypatch.setenv("PRAG_SHADOW_AFFINE", mode)
        ix = pra.HipFlatIndex(d, "l2", "f16")           # capacity grows: the shadow is rebuilt from row 0 on the way
        monkeypatch.delenv("PRAG_SHADOW_AFFINE")
        ix.set_shadow(2)
        ix.add(X[: N // 2])
        ix.add(X[N // 2:])
        ix.add(torch.from_numpy(onp.synth_rows(3, 0, 5000, d)).cuda() * 0.05)     # iid rows: not what the map was fitted on
        res[mode] = ix.search(q, k)
        surv[mode] = ix.last_survivors()["per_query"]
        if mode == "1":
            stored = ix.reconstruct_n(0, ix.ntotal)
            D0, I0 = oracle_c.flat_search(stored, q.cpu().numpy(), k, onp.METRIC_L2)
            _check(res[mode][0].cpu().numpy(), res[mode][1].cpu().numpy(), D0, I0, onp.METRIC_L2)
        ix.close()
    assert torch.equal(res["1"][1], res["0"][1]) and torch.equal(res["1"][0], res["0"][0])
    assert torch.equal(res["1"][1], res["2"][1]) and torch.equal(res["1"][0], res["2"][0])
    assert surv["1"] * 3 < surv["0"], surv          # rows and queries centred on the column means: several times tighter


def test_work_beside_the_tail_of_a_search():
    """prag_index_stream_wait_scan: a second stream waits for the corpus scan of the latest search only, so the gate of
    the next batch can run beside the bound kernel / rerank.  Results of both are what one stream gives, eagerly and
    replayed from a captured graph (the wait becomes a fork edge); the first call only switches the recording on."""
    import torch
    import probing_rag_amd as pra
    from tests.golden import cases
    N, d, k = 300_000, 768, 10
    ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
    ix.add_synthetic(42, 0, N)
    ix.set_shadow(2)
    ix.prepare()
    q = torch.from_numpy(onp.synth_rows(7, 0, 64, d)).cuda()
    case = dict(cases.PROBER_CASES[1], B=96)
    ens = pra.HipProberEnsemble(case["L"], case["d"], 2, weights="f16")
    for l in range(case["L"]):
        ens.load_layer(l, cases.synth_state(case["wseed"] + l, case["d"]))
    x = torch.from_numpy(cases.synth_x(case["xseed"], case["L"], 96, case["d"], 1.0)).cuda().half()
    D0, I0 = ix.search(q, k)
    want = [t.clone() for t in ens.gate(x, 0, 0.0)]
    out = (torch.empty_like(D0), torch.empty_like(I0))
    gout = tuple(torch.empty_like(t) for t in want)
    side = torch.cuda.Stream()

    def one_pass():
        ix.search(q, k, out=out)
        ix.stream_wait_scan(side)
        with torch.cuda.stream(side):
            ens.gate(x, 0, 0.0, out=gout)
        torch.cuda.current_stream().wait_stream(side)

    ix.stream_wait_scan(side)             # recording on; nothing to wait for
    for _ in range(5):
        for t in (*out, *gout):
            t.zero_()
        one_pass()
        torch.cuda.synchronize()
        assert torch.equal(out[1], I0) and torch.equal(out[0], D0)
        assert all(torch.equal(a, b) for a, b in zip(gout, want))
    # direct scans record the event too (behind their scan + rerank launches)
    ix.set_shadow(0)
    one_pass()
    torch.cuda.synchronize()
    assert torch.equal(out[1], I0)
    ix.set_shadow(2)
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        one_pass()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cs):
            one_pass()
    for _ in range(3):
        for t in (*out, *gout):
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[1], I0) and all(torch.equal(a, b) for a, b in zip(gout, want))


def test_flagged_queries_are_retried_32_at_a_time_before_the_exact_scan(monkeypatch):
    """Round 5 retry tier: on rows with outlier coordinates and the round 2-4 shadow (PRAG_SHADOW_AFFINE=0) the 64-query
    tiles - one int8 query term - overflow their candidate regions for nearly every query; the 32-query tiles - both
    terms - do not.  With the tier (armed by the first search's flag count, or forced) the flagged queries are searched
    again 32 at a time and hardly any reaches the exact float64 scan; results are the definition's either way, for
    device and host i/o, and the tier disarms nothing it should not (a clean corpus stays unarmed)."""
    import torch
    import probing_rag_amd as pra
    from probing_rag_amd.synth import embedding_like_rows, embedding_structure
    N, d, k = 300_000, 768, 10
    st = embedding_structure(11, d)
    X = embedding_like_rows(11, 0, N, d, structure=st)
    q = embedding_like_rows(1011, 0, 64, d, structure=st)
    monkeypatch.setenv("PRAG_SHADOW_AFFINE", "0")
    counts = {}
    for mode in ("0", "1", "auto"):
        if mode == "auto":
            monkeypatch.delenv("PRAG_RETRY_TIER", raising=False)
        else:
            monkeypatch.setenv("PRAG_RETRY_TIER", mode)
        ix = pra.HipFlatIndex(d, "l2", "f16", capacity=N)
        ix.set_adaptive(True)       # (arming by history is the adaptive plan's: also under PRAG_ADAPTIVE=0)
        ix.set_shadow(2)
        ix.add(X)
        if mode == "0":
            stored = ix.reconstruct_n(0, N)
            D0, I0 = oracle_c.flat_search(stored, q.cpu().numpy(), k, onp.METRIC_L2)
        D, I = ix.search(q, k)
        _check(D.cpu().numpy(), I.cpu().numpy(), D0, I0, onp.METRIC_L2)
        counts[mode] = [ix.last_exact_fallbacks()]
        torch.cuda.synchronize()
        D, I = ix.search(q, k)                      # (adaptive: armed by the first search's count)
        _check(D.cpu().numpy(), I.cpu().numpy(), D0, I0, onp.METRIC_L2)
        counts[mode].append(ix.last_exact_fallbacks())
        Dn, In = ix.search(q.cpu().numpy(), k)      # host i/o: decided on the count that came back with the results
        _check(Dn, In, D0, I0, onp.METRIC_L2)
        counts[mode].append(ix.last_exact_fallbacks())
        # 70 queries: three parts of 32 (the last one padded)
        q70 = embedding_like_rows(1012, 0, 70, d, structure=st)
        D7, I7 = ix.search(q70, k)
        D70, I70 = oracle_c.flat_search(stored, q70.cpu().numpy(), k, onp.METRIC_L2)
        _check(D7.cpu().numpy(), I7.cpu().numpy(), D70, I70, onp.METRIC_L2)
        ix.close()
    assert counts["0"][0] > 32 and counts["0"][1] > 32, counts             # the cliff: most queries in the exact scan
    assert max(counts["1"]) <= 8, counts                                    # forced: the two-term tiles clear them
    assert counts["auto"][0] > 32 and counts["auto"][1] <= 8 and counts["auto"][2] <= 8, counts
    # a clean corpus never arms the tier (and pays nothing for it)
    monkeypatch.delenv("PRAG_SHADOW_AFFINE")
    ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
    ix.set_shadow(2)
    ix.add_synthetic(42, 0, N)
    qq = torch.from_numpy(onp.synth_rows(7, 0, 64, d)).cuda()
    for _ in range(3):
        ix.search(qq, k)
        torch.cuda.synchronize()
        assert ix.last_exact_fallbacks() == 0
    ix.close()


@pytest.mark.parametrize("store,scan_gate", [("f16", "0"), ("f32", "0"), ("f16", "1"), ("f32", "1")])
def test_search_and_gate_equals_the_two_calls(store, scan_gate, monkeypatch):
    """prag_search_and_gate: the top-k of a query batch AND the gate over the next batch of pooled states in one call; on
    a two-level search the prober's workgroups ride in the launch of the search's bound kernel (bound_gate_kernel: the
    bodies of shadow_bound_kernel and prober16_kernel side by side; PRAG_SCAN_GATE=0) or behind the scan's workgroups
    in the SCAN's launch (scan8_gate_kernel: 64-query tiles; forced here with PRAG_SCAN_GATE=1 - by itself the library
    takes it when the gate fits under the scan).  Whatever path it takes - fused launch at every prober tile height,
    direct scan, float32 states, the small-batch gate, no gate at all, a captured graph - D, I, logits, sums and
    decisions are those of index.search and ens.gate."""
    import torch
    import probing_rag_amd as pra
    from tests.golden import cases
    monkeypatch.setenv("PRAG_SCAN_GATE", scan_gate)
    if scan_gate == "1":        # ... with the exact-bound kernel on at this size: its launch then finishes the gate (bound_finish_kernel)
        monkeypatch.setenv("PRAG_SHADOW_BOUND", "1")
    N, d, k = 300_000, 768, 10
    ix = pra.HipFlatIndex(d, "cos", store, capacity=N)
    ix.add_synthetic(42, 0, N)
    ix.set_shadow(2)
    ix.prepare()
    q = torch.from_numpy(onp.synth_rows(7, 0, 64, d)).cuda()
    case = cases.PROBER_CASES[1]
    ens = pra.HipProberEnsemble(case["L"], case["d"], 2, weights="f16")
    for l in range(case["L"]):
        ens.load_layer(l, cases.synth_state(case["wseed"] + l, case["d"]))
    D0, I0 = ix.search(q, k)
    for Bg in (40, 96, 512, 1400, 4096):                 # 32-, 64- and 128-row prober tiles, ragged last tiles
        x = torch.from_numpy(cases.synth_x(case["xseed"] + Bg, case["L"], Bg, case["d"], 1.0)).cuda().half()
        want = [t.clone() for t in ens.gate(x, 1, 0.25)]
        for nq in (64, 1, 33):                           # 64- and 32-query tiles of the scan
            (D, I), got = pra.search_and_gate(ix, q[:nq], k, ens, x, 1, 0.25)
            assert ix.last_plan()["family"] == "scan8_kernel"
            assert torch.equal(I, I0[:nq]) and torch.equal(D, D0[:nq])
            assert all(torch.equal(a, b) for a, b in zip(got, want)), (Bg, nq)
    x = torch.from_numpy(cases.synth_x(case["xseed"], case["L"], 96, case["d"], 1.0)).cuda()
    for xx in (x, x.half()[:, :1].contiguous()):         # float32 states; ONE pooled state (the small-batch gate)
        want = [t.clone() for t in ens.gate(xx, 0, 0.0)]
        (D, I), got = pra.search_and_gate(ix, q, k, ens, xx)
        assert torch.equal(I, I0) and all(torch.equal(a, b) for a, b in zip(got, want))
    xh = x.half()
    want = [t.clone() for t in ens.gate(xh, 0, 0.0)]
    ix.set_shadow(0)                                     # direct scan: search, then the gate
    (D, I), got = pra.search_and_gate(ix, q, k, ens, xh)
    assert torch.equal(I, I0) and all(torch.equal(a, b) for a, b in zip(got, want))
    ix.set_shadow(2)
    # captured into a graph and replayed
    out = (torch.empty_like(D0), torch.empty_like(I0))
    gout = tuple(torch.empty_like(t) for t in want)
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        pra.search_and_gate(ix, q, k, ens, xh, out=out, gate_out=gout)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cs):
            pra.search_and_gate(ix, q, k, ens, xh, out=out, gate_out=gout)
    for _ in range(3):
        for t in (*out, *gout):
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[1], I0) and all(torch.equal(a, b) for a, b in zip(gout, want))


def test_gather_launch_is_skipped_only_while_nothing_needs_it(monkeypatch):
    """Round 5: behind the exact-bound kernel the sliced gather is ~9 us of nothing when the bound kernel finished every
    query; after 16 such searches the launch is dropped.  A query the bound kernel then cannot finish - here 400 copies of
    one row tie under its bound, more than the 256 it scores itself - goes through the flag list (retry tier / exact
    scan) and re-arms the gather: the results are the definition's before, during and after."""
    import torch
    import probing_rag_amd as pra
    monkeypatch.setenv("PRAG_SHADOW_BOUND", "1")
    N, d, k = 120_000, 768, 10
    X = onp.synth_rows(42, 0, N, d)
    rng = np.random.default_rng(3)
    where = rng.choice(N, 400, replace=False)
    X[where] = X[where[0]]
    ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
    ix.set_shadow(2)
    ix.add(X)
    stored = ix.reconstruct_n(0, N)
    q = torch.from_numpy(onp.synth_rows(7, 0, 64, d)).cuda()
    D0, I0 = oracle_c.flat_search(stored, q.cpu().numpy(), k, onp.METRIC_COS)
    for _ in range(24):                              # clean searches: the gather is disarmed on the way
        D, I = ix.search(q, k)
        torch.cuda.synchronize()
    _check(D.cpu().numpy(), I.cpu().numpy(), D0, I0, onp.METRIC_COS)
    assert ix.last_exact_fallbacks() == 0
    qh = q.clone()
    qh[5] = torch.from_numpy(X[where[0]]).cuda()      # 400 exact ties at the top of this query
    Dh0, Ih0 = oracle_c.flat_search(stored, qh.cpu().numpy(), k, onp.METRIC_COS)
    for _ in range(4):                               # the first one meets a skipped gather, the next ones an armed one
        D, I = ix.search(qh, k)
        torch.cuda.synchronize()
        _check(D.cpu().numpy(), I.cpu().numpy(), Dh0, Ih0, onp.METRIC_COS)
    assert sorted(I[5].tolist()) == sorted(np.sort(where)[:k].tolist())      # ties -> lowest ids
    D, I = ix.search(q, k)
    _check(D.cpu().numpy(), I.cpu().numpy(), D0, I0, onp.METRIC_COS)
    ix.close()


def test_scan_workgroups_are_measured_not_assumed(monkeypatch):
    """Round 5: a two-level scan of <= 64 queries runs on 7/8 of the CUs or on all of them - iid rows are 1.6-3 % faster
    on 7/8 (fewer concurrent HBM streams), embedding-shaped rows 5-8 % slower - so an index times eight of its own
    searches (alternating, events never waited for) and keeps the faster.  Results never depend on the grid; the plan on
    record names the grid that ran; PRAG_SCAN_WG_TUNE pins either choice; a caller's own cap is kept."""
    import torch
    import probing_rag_amd as pra
    N, d, k = 1_200_000, 768, 10
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count

    def prime_le(n):              # the library's own grids are the largest primes not above 7/8 of the CUs / all of them
        return next(c for c in range(n, 1, -1) if all(c % f for f in range(2, int(c ** 0.5) + 1)))
    g78, gall = prime_le(n_cu * 7 // 8), prime_le(n_cu)
    q = torch.from_numpy(onp.synth_rows(7, 0, 64, d)).cuda()

    def build():
        ix = pra.HipFlatIndex(d, "cos", "f16", capacity=N)
        ix.set_adaptive(True)       # (the measurement is the adaptive plan's: also under PRAG_ADAPTIVE=0)
        ix.add_synthetic(42, 0, N)
        ix.set_shadow(2)
        ix.prepare()
        return ix

    ix = build()
    D0, I0 = ix.search(q, k)
    grids = []
    for _ in range(14):
        D, I = ix.search(q, k)
        torch.cuda.synchronize()
        assert torch.equal(I, I0) and torch.equal(D, D0)
        grids.append(ix.last_plan()["grid"])
    assert set(grids[:8]) == {g78, gall}                    # both were tried ...
    assert len(set(grids[-4:])) == 1 and grids[-1] in (g78, gall)   # ... and one was kept
    D1, I1 = ix.search(q[:1], k)                                # <= 32 queries: measured separately
    assert ix.last_plan()["grid"] in (g78, gall)
    ix.set_scan_workgroups(100)                                 # the caller's own cap
    D, I = ix.search(q, k)
    assert ix.last_plan()["grid"] == 100 and torch.equal(I, I0)
    ix.close()
    for mode, want in (("0", g78), ("1", gall)):
        monkeypatch.setenv("PRAG_SCAN_WG_TUNE", mode)
        ix = build()
        for _ in range(3):
            D, I = ix.search(q, k)
            assert ix.last_plan()["grid"] == want and torch.equal(I, I0)
        ix.close()
