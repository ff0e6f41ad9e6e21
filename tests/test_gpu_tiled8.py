"""GPU: batches of > 128 queries on an index that keeps the 8-bit shadow (flat_mm.hip, I8 kernels).
First tier: candidate selection on int8 matrix tiles over the shadow (256 candidates per query, the
shadow's Cauchy-Schwarz bound in the certificate); a batch in which any query fails that certificate is
repeated on the fp16 tiles.  Either way the results must be the float64 definition's, bit for bit
(reference call: utils.py:378-380 batch_topk_sim -> IndexFlat.search)."""
import numpy as np
import pytest

from oracle import oracle_c, oracle_np as onp

pytestmark = pytest.mark.gpu

METRICS = [onp.METRIC_L2, onp.METRIC_IP, onp.METRIC_COS]


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
@pytest.mark.parametrize("N,B,k,d", [(30_000, 300, 10, 768), (2049, 129, 5, 768), (70_000, 257, 10, 512),
                                     (9_000, 513, 26, 1024), (300, 200, 10, 768), (1, 130, 5, 768),
                                     (12_000, 1100, 10, 768)])      # two 1024-query chunks
def test_int8_tiles_match_definition(metric, store, N, B, k, d):
    import probing_rag_amd as pra
    X = onp.synth_rows(42, 0, N, d)
    if N > 40:
        X[N // 2] = X[3]
        X[N - 1] = X[3]              # exact duplicates: ties resolved by id
    Q = onp.synth_rows(7, 0, B, d)
    if N > 40:
        Q[0] = X[3]
    ix = pra.HipFlatIndex(d, metric, store)
    ix.set_shadow(2)
    ix.add(X[: N // 2])
    ix.add(X[N // 2:])
    D, I = ix.search(Q, k)
    assert ix.last_tiled8() >= 0                     # the int8 tiles ran
    D0, I0 = oracle_c.flat_search(_stored(X, metric, store), Q, k, metric)
    _check(D, I, D0, I0, metric)
    # the same through device tensors (the tier decision is taken on the device either way)
    import torch
    Dt, It = ix.search(torch.from_numpy(Q).cuda(), k)
    assert ix.last_tiled8() >= 0
    _check(Dt.cpu().numpy(), It.cpu().numpy(), D0, I0, metric)
    # shadow off: the fp16 tiles alone answer, same results
    ix.set_shadow(0)
    D1, I1 = ix.search(Q, k)
    assert ix.last_tiled8() == -1
    _check(D1, I1, D0, I0, metric)
    ix.close()


@pytest.mark.parametrize("metric", METRICS)
def test_first_tier_answers_a_random_corpus(metric):
    """Exchangeable rows: every query clears the 8-bit certificate with 256 candidates (no second tier,
    no exact fallback) and the ids are those of the fp16 tiles."""
    import probing_rag_amd as pra
    import torch
    N, d, B, k = 1_200_000, 768, 1000, 10
    ix = pra.HipFlatIndex(d, metric, "f16", capacity=N)
    ix.add_synthetic(42, 0, N)
    q = torch.from_numpy(onp.synth_rows(7, 0, B, d)).cuda()
    ix.set_shadow(0)
    D0, I0 = ix.search(q, k)
    assert ix.last_tiled8() == -1
    ix.set_shadow(2)           # (mode 1 takes the int8 tiles from 2 Mi rows on: below that the fp16 tiles are faster)
    ix.prepare()
    D1, I1 = ix.search(q, k)
    assert ix.last_tiled8() == 0, ix.last_tiled8()
    assert ix.last_exact_fallbacks() == 0
    assert torch.equal(I0, I1)
    assert torch.equal(D0, D1)
    # a sample of the queries against the float64 oracle over the first 150 k rows (searched separately)
    sub = pra.HipFlatIndex(d, metric, "f16", capacity=150_000)
    sub.set_shadow(2)
    sub.add_synthetic(42, 0, 150_000)
    Ds, Is = sub.search(q[:140], k)
    assert sub.last_tiled8() >= 0
    _, Iref = oracle_c.flat_search(sub.reconstruct_n(0, 150_000), q[:140].cpu().numpy(), k, metric)
    assert np.array_equal(Is.cpu().numpy(), Iref)
    sub.close()
    ix.close()


@pytest.mark.parametrize("store", ["f16", "f32"])
@pytest.mark.parametrize("metric", METRICS)
def test_second_tier_on_a_corpus_of_look_alikes(metric, store, monkeypatch):
    """Thousands of rows inside the 8-bit error band of every query's k-th score: 256 candidates cannot clear
    the certificate, the batch is repeated on the fp16 tiles (and their exact fallback) - results still the
    definition's, through host arrays and through device tensors.  (Rows = one base vector + noise: with the shadow
    centred on the column means - round 5 - the int8 grid resolves the noise itself and the first tier answers such a
    corpus; the second tier is exercised on the round 2-4 shadow, PRAG_SHADOW_AFFINE=0, and the default is checked for
    the definition's results next to it.)"""
    import probing_rag_amd as pra
    import torch
    N, d, B, k = 20_000, 768, 150, 10
    rng = np.random.default_rng(5)
    base = onp.synth_rows(5, 0, 1, d)[0]
    X = (base[None, :] + 2e-3 * rng.standard_normal((N, d))).astype(np.float32)
    Q = (base[None, :] + 2e-3 * rng.standard_normal((B, d))).astype(np.float32)
    D0, I0 = oracle_c.flat_search(_stored(X, metric, store), Q, k, metric)
    ixc = pra.HipFlatIndex(d, metric, store)          # the default (centred) shadow: whichever tier answers, exact
    ixc.set_shadow(2)
    ixc.add(X)
    Dc, Ic = ixc.search(Q, k)
    assert ixc.last_tiled8() >= 0
    _check(Dc, Ic, D0, I0, metric)
    ixc.close()
    monkeypatch.setenv("PRAG_SHADOW_AFFINE", "0")     # read when the index is created
    ix = pra.HipFlatIndex(d, metric, store)
    monkeypatch.delenv("PRAG_SHADOW_AFFINE")
    ix.set_adaptive(True)        # (the auto-off of the int8 tiles is the adaptive plan's: also under PRAG_ADAPTIVE=0)
    ix.set_shadow(2)
    ix.add(X)
    D, I = ix.search(Q, k)
    assert ix.last_tiled8() > 0
    _check(D, I, D0, I0, metric)
    Dt, It = ix.search(torch.from_numpy(Q).cuda(), k)
    assert ix.last_tiled8() > 0
    _check(Dt.cpu().numpy(), It.cpu().numpy(), D0, I0, metric)
    # two whole-batch repeats in a row: the index stops trying the int8 tiles (every search would pay for both
    # tiers) until rows are added, set_shadow is called, or 64 eligible searches have gone by (one probe, then
    # 128, ... 4096)
    D3, I3 = ix.search(Q, k)
    assert ix.last_tiled8() == -2
    _check(D3, I3, D0, I0, metric)
    ix.set_shadow(2)
    D4, I4 = ix.search(Q, k)
    assert ix.last_tiled8() > 0
    _check(D4, I4, D0, I0, metric)
    ix.search(Q, k)
    assert ix.last_tiled8() > 0                      # second repeat in a row: off again
    qd = torch.from_numpy(Q).cuda()
    seen = []
    for _ in range(64):
        ix.search(qd, k)
        seen.append(ix.last_tiled8())
    assert seen[:63] == [-2] * 63 and seen[63] > 0   # the probe
    ix.search(qd, k)
    assert ix.last_tiled8() == -2                    # one failed probe switches them off again, for longer
    ix.close()


@pytest.mark.parametrize("store", ["f16", "f32"])
@pytest.mark.parametrize("metric", METRICS)
def test_a_few_failed_queries_are_searched_again_on_their_own(metric, store):
    """A random corpus with ONE cluster of look-alikes and three queries inside it: only those fail the 8-bit
    certificate; they are gathered into a compact batch, searched by the <= 128-query kernels and scattered back
    (the other queries keep the first tier's results) - the definition's results for everybody."""
    import probing_rag_amd as pra
    import torch
    N, d, B, k = 40_000, 768, 203, 10
    rng = np.random.default_rng(9)
    X = onp.synth_rows(42, 0, N, d)
    base = onp.synth_rows(5, 0, 1, d)[0] * np.float32(1.7)
    where = rng.choice(N, 700, replace=False)
    X[where] = (base[None, :] + 2e-3 * rng.standard_normal((700, d))).astype(np.float32)
    Q = onp.synth_rows(7, 0, B, d)
    inside = [5, 100, 202]
    for i in inside:
        Q[i] = (base + 2e-3 * rng.standard_normal(d)).astype(np.float32)
    ix = pra.HipFlatIndex(d, metric, store)
    ix.set_shadow(2)
    ix.add(X)
    D0, I0 = oracle_c.flat_search(_stored(X, metric, store), Q, k, metric)
    D, I = ix.search(Q, k)
    n_failed = ix.last_tiled8()
    assert 1 <= n_failed <= 50, n_failed         # (the three in the cluster; a random query may sit near it too)
    _check(D, I, D0, I0, metric)
    Dt, It = ix.search(torch.from_numpy(Q).cuda(), k)
    assert 1 <= ix.last_tiled8() <= 50
    _check(Dt.cpu().numpy(), It.cpu().numpy(), D0, I0, metric)
    ix.close()


def test_sharded_large_batch_equals_unsharded():
    """Row shards searched with tagged ids through the int8 tiles, merged: identical to one index."""
    import probing_rag_amd as pra
    import torch
    N, d, B, k = 90_000, 768, 260, 10
    X = onp.synth_rows(42, 0, N, d)
    Q = torch.from_numpy(onp.synth_rows(7, 0, B, d)).cuda()
    one = pra.HipFlatIndex(d, "cos", "f16")
    one.set_shadow(2)
    one.add(X)
    D0, I0 = one.search(Q, k)
    assert one.last_tiled8() >= 0
    from probing_rag_amd.sharded import search_shards_on_one_gpu
    shards = []
    bounds = [0, 20_001, 55_555, N]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        sh = pra.HipFlatIndex(d, "cos", "f16")
        sh.set_shadow(2)
        sh.add(X[lo:hi])
        shards.append(sh)
    D1, I1 = search_shards_on_one_gpu(shards, Q, k, "cos")
    assert all(sh.last_tiled8() >= 0 for sh in shards)
    assert torch.equal(I0, I1)
    assert torch.allclose(D0, D1, atol=1e-6)
    for sh in shards:
        sh.close()
    one.close()


def test_large_batch_search_is_graph_capturable_and_never_waits(monkeypatch):
    """The device-io search of > 128 queries on the int8 tiles issues a fixed list of launches whatever the data:
    both second-tier continuations are always enqueued and switched by the failed count on the device.  So it can be
    captured into a HIP graph and replayed - on a corpus where the first tier answers, where a few queries fail
    (compact batch) and where everything fails (whole batch on the fp16 tiles) - with the definition's results."""
    import probing_rag_amd as pra
    import torch
    d, k = 768, 10
    rng = np.random.default_rng(9)
    base = onp.synth_rows(5, 0, 1, d)[0] * np.float32(1.7)

    def corpus(kind, N, B):
        X = onp.synth_rows(42, 0, N, d)
        Q = onp.synth_rows(7, 0, B, d)
        if kind == "few":
            where = rng.choice(N, 700, replace=False)
            X[where] = (base[None, :] + 2e-3 * rng.standard_normal((700, d))).astype(np.float32)
            for i in (5, 100, B - 1):
                Q[i] = (base + 2e-3 * rng.standard_normal(d)).astype(np.float32)
        elif kind == "all":
            X = (base[None, :] + 2e-3 * rng.standard_normal((N, d))).astype(np.float32)
            Q = (base[None, :] + 2e-3 * rng.standard_normal((B, d))).astype(np.float32)
        return X, Q

    for kind, N, B in (("none", 30_000, 300), ("few", 40_000, 203), ("all", 20_000, 150)):
        X, Q = corpus(kind, N, B)
        if kind == "all":       # (one base vector + noise: the centred shadow of round 5 answers it on the first tier;
            monkeypatch.setenv("PRAG_SHADOW_AFFINE", "0")     #  the whole-batch repeat is exercised on the round 2-4 shadow)
        ix = pra.HipFlatIndex(d, "ip", "f16")
        monkeypatch.delenv("PRAG_SHADOW_AFFINE", raising=False)
        ix.set_shadow(2)
        ix.add(X)
        D0, I0 = oracle_c.flat_search(_stored(X, onp.METRIC_IP, "f16"), Q, k, onp.METRIC_IP)
        qd = torch.from_numpy(Q).cuda()
        out = (torch.empty((B, k), dtype=torch.float32, device="cuda"), torch.empty((B, k), dtype=torch.int64, device="cuda"))
        ix.search(qd, k, out=out)                       # sizes every workspace (allocation is not capturable)
        n_failed = ix.last_tiled8()
        assert (n_failed == 0) if kind == "none" else (1 <= n_failed <= 50) if kind == "few" else n_failed > B // 4, (kind, n_failed)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                ix.search(qd, k, out=out)
        assert ix.last_tiled8() == -3                   # a captured search reports nothing back
        for _ in range(2):
            out[0].zero_()
            out[1].zero_()
            g.replay()
            torch.cuda.synchronize()
            _check(out[0].cpu().numpy(), out[1].cpu().numpy(), D0, I0, onp.METRIC_IP)
        ix.set_shadow(2)                                # (the 'all' corpus: re-arm the tiles for the next handle state)
        del g
        ix.close()
