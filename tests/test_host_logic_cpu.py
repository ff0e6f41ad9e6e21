"""CPU: host-side logic that needs no kernels — loop control flow, evidence
formatting, row partitioning, faiss file header, reference-shaped helpers."""
import struct

import numpy as np

import probing_rag_amd as pra
from oracle import oracle_np as onp


def test_partition_rows_covers_everything_contiguously():
    for n, w in ((21_000_000, 8), (10, 3), (5, 8), (0, 4), (1000, 1)):
        spans = [pra.partition_rows(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and a <= b and c <= d
    assert pra.partition_rows(21_000_000, 8, 0) == (0, 2_625_000)


def test_return_evidences_format():
    # exp_rag.py:369-379
    assert pra.return_evidences(["a", "b", "c"]) == "passage 1: a\npassage 2: b\npassage 3: c"
    assert pra.return_evidences([]) == ""


def _drive(decisions):
    """Run retrieve_decide with a scripted gate; returns (pred, retr_count, log)."""
    it = iter(decisions)
    log = {"queries": [], "gen": 0, "resets": 0}

    def generate(inp):
        log["gen"] += 1
        return f"out{log['gen']}"

    def retrieve(text, k):
        log["queries"].append((text, k))
        return np.zeros((1, k), np.float32), np.arange(k, dtype=np.int64)[None, :]

    pred, rc = pra.retrieve_decide(
        "Q?", "ids0", generate=generate, gate=lambda: next(it), retrieve=retrieve,
        lookup=lambda ids: [f"doc{i}" for i in ids], make_prompt=lambda q, ev: f"{ev}|{q}",
        tokenize=lambda s: s, to_string=lambda o: [f"text({o})"],
        reset=lambda: log.__setitem__("resets", log["resets"] + 1), k=5)
    return pred, rc, log


def test_retrieve_decide_matches_reference_cap_semantics():
    # exp_rag.py:417-468: <= 4 rounds, retr_count saturates at 3, round >= 2 queries
    # with the full decoded text (exp_rag.py:435, 457)
    for decisions in ([0], [1, 0], [1, 1, 0], [1, 1, 1, 0], [1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1]):
        pred, rc, log = _drive(decisions)
        want_rc, want_rounds = onp.retr_count_from_decisions(decisions)
        assert rc == want_rc and len(log["queries"]) == want_rounds
        assert log["gen"] == 1 + want_rounds and log["resets"] == 1 + want_rounds
        if want_rounds:
            assert log["queries"][0] == ("Q?", 5)
            for j, (text, _) in enumerate(log["queries"][1:], start=2):
                assert text == f"text(out{j})"
            assert pred == f"text(out{1 + want_rounds})"
        else:
            assert pred == "text(out1)"


def test_reference_shaped_helpers():
    class Cfg:
        d_model, tokenizer_name = 2048, "google/gemma-2b"

    class Model:
        cfg = Cfg()

    cfgs = pra.load_prober_cfg_gemma_2b(Model(), pra.Config_Maker, "resid_post", "cuda", 6, 17, 2)
    assert [c.layer for c in cfgs] == [6, 8, 10, 12, 14, 16]       # exp_rag.py:311
    assert all(c.method == "tokens_mean" and c.num_classes == 2 and c.d_model == 2048 for c in cfgs)

    class T:
        def __init__(self, v): self.v = v
        def to(self, dev): return ("cpu", self.v)

    got = pra.return_prober_logit_gemma_2b(lambda cfg, m: T(m(cfg)), [1, 2], [lambda c: c * 10, lambda c: c * 100])
    assert got == [("cpu", 10), ("cpu", 200)]

    class Enc:
        def encode(self, q): return np.full((len(q), 4), 2.0, np.float32)

    class Ix:
        def search(self, x, k): return x[:, :k], np.zeros((len(x), k), np.int64)

    D, I = pra.batch_topk_sim(Enc(), ["a", "b"], Ix(), k=3)             # utils.py:378-380
    assert D.shape == (2, 3) and I.shape == (2, 3)


def test_write_index_emits_faiss_flat_header(tmp_path):
    class Fake:
        d, ntotal, metric = 4, 3, pra.index.METRIC_L2
        def reconstruct_n(self, a, n): return np.arange(12, dtype=np.float32).reshape(3, 4)

    p = tmp_path / "x.bin"
    pra.write_index(Fake(), str(p))
    raw = p.read_bytes()
    assert raw[:4] == b"IxF2"
    d, nt, _, _, trained, metric = struct.unpack("<iqqqBi", raw[4:4 + 33])
    assert (d, nt, trained, metric) == (4, 3, 1, 1)
    (nf,) = struct.unpack("<Q", raw[37:45])
    assert nf == 12 and np.frombuffer(raw[45:], np.float32).tolist() == list(range(12))


def test_merge_topk_properties():
    """Exchange-step semantics on random shard results (hypothesis): merging the shards' sorted
    top-k lists equals the top-k of the concatenation under (score, id) order, for both metric
    directions, with padded (-1) entries and ties across shards."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 5), st.integers(1, 4), st.integers(1, 6), st.integers(0, 2 ** 31 - 1),
           st.sampled_from([onp.METRIC_L2, onp.METRIC_IP]))
    def check(parts, B, k, seed, metric):
        rng = np.random.default_rng(seed)
        Ds, Is = [], []
        next_id = 0
        for _ in range(parts):
            n = int(rng.integers(0, 2 * k + 1))              # rows in this shard (may be < k)
            ids = np.arange(next_id, next_id + n)
            next_id += n
            sc = rng.integers(0, 4, size=(B, n)).astype(np.float32)   # few distinct values -> ties
            D = np.full((B, k), np.finfo(np.float32).max if metric == onp.METRIC_L2 else -np.finfo(np.float32).max, np.float32)
            I = np.full((B, k), -1, np.int64)
            for b in range(B):
                key = sc[b] if metric == onp.METRIC_L2 else -sc[b]
                o = np.lexsort((ids, key))[:k]
                D[b, :len(o)] = sc[b, o]
                I[b, :len(o)] = ids[o]
            Ds.append(D)
            Is.append(I)
        D, I = onp.merge_topk(Ds, Is, k, metric)
        allD, allI = np.concatenate(Ds, 1), np.concatenate(Is, 1)
        for b in range(B):
            valid = allI[b] >= 0
            key = allD[b, valid] if metric == onp.METRIC_L2 else -allD[b, valid]
            o = np.lexsort((allI[b, valid], key))[:k]
            want = allI[b, valid][o]
            assert I[b, :len(want)].tolist() == want.tolist()
            assert (I[b, len(want):] == -1).all()

    check()


def _cfg(model_id="google/gemma-2b", layer=6, position="resid_post", method="tokens_mean"):
    class Cfg:
        pass
    c = Cfg()
    c.model_id, c.layer, c.position, c.method, c.device, c.d_model, c.num_classes = \
        model_id, layer, position, method, "cuda", 2048, 2
    return c


def test_checkpoint_path_table_matches_the_reference():
    """utils.py:303-326, every branch: the strings below are what the reference passes to torch.load
    for --ds in {25,50,75,777,3,333,366,3000,1000, anything else} and for the Mistral model id."""
    c = _cfg(layer=12, position="resid_mid")
    want = {
        25: "ckpt/_25/0.25_gemma-2b_tokens_mean_2_l12_resid_mid_ep1.pt",
        50: "ckpt/_5/0.5_gemma-2b_tokens_mean_2_l12_resid_mid_ep1.pt",
        75: "ckpt/_75/0.75_gemma-2b_tokens_mean_2_l12_resid_mid_ep1.pt",
        777: "ckpt/_75_full/0.75_gemma-2b_tokens_mean_2_l12_resid_mid_ep.pt",
        3: "ckpt/_3/in3_1.0_gemma-2b_tokens_mean_2_l12_resid_mid_ep1.pt",
        333: "ckpt/_3_3/in3_0.33_gemma-2b_tokens_mean_2_l12_resid_mid_ep.pt",
        366: "ckpt/_3_6/in3_0.66_gemma-2b_tokens_mean_2_l12_resid_mid_ep.pt",
        3000: "ckpt/_3_1000/in3_1000_gemma-2b_tokens_mean_2_l12_resid_mid_ep11.pt",
        1000: "ckpt/_1000/1000_gemma-2b_tokens_mean_2_l12_resid_mid_ep11.pt",
        0: "ckpt/prob_model_cot_v1/gemma-2b_linear995_tokens_mean_probe_2_l12_resid_mid_1.pt",
        42: "ckpt/prob_model_cot_v1/gemma-2b_linear995_tokens_mean_probe_2_l12_resid_mid_1.pt",
    }
    for ds, path in want.items():
        assert pra.prober_checkpoint_path(ds, c) == path
    m = _cfg("mistralai/Mistral-7B-Instruct-v0.1", 14, "attn_out")
    assert pra.prober_checkpoint_path(3, m) == \
        "ckpt/probing_ckpt/Mistral-7B-Instruct-v0.1_tokens_mean_probe_2_l14_attn_out_1.pt"
    # any other model id: the reference's `assert '<string>'` is a no-op and nothing is loaded
    assert pra.prober_checkpoint_path(3, _cfg("meta-llama/Llama-2-7b")) is None


def test_docstore_round_trip_and_lookup(tmp_path):
    """make_indexer.py:461-464 writer, exp_rag.py:298 reader, exp_rag.py:436 lookup."""
    import pandas as pd
    texts = ["first passage, with a comma", 'second "quoted" passage', "third\nline break", "4th"]
    ids = ["a1", "b2", "c3", "d4"]
    p = str(tmp_path / "nq_index_2.csv")
    pra.write_docstore(texts, ids, p)
    assert open(p).readline().strip() == "doc,doc_id"                   # df.columns = ['doc','doc_id'], index=False
    ref = pd.DataFrame([texts, ids]).T                                   # the reference's own three lines
    ref.columns = ["doc", "doc_id"]
    corpus = pra.read_docstore(p)
    assert corpus.equals(pd.read_csv(p)) and list(corpus["doc"]) == texts and list(corpus["doc_id"]) == ids
    I = np.array([[2, 0, 3, 1, 2]], dtype=np.int64)
    assert pra.lookup_passages(corpus, I[0].tolist()) == list(ref.iloc[I[0].tolist(), 0])
    look = pra.Docstore(p)
    assert len(look) == 4 and look([3, 3, 0]) == ["4th", "4th", texts[0]]
    with np.testing.assert_raises(IndexError):
        look([4])


def _faiss_flat_bytes(fourcc, d, ntotal, metric_type, rows, n_floats=None):
    """An IndexFlat file assembled by hand from the layout of faiss's index_write.cpp
    (write_index_header + WRITEXBVECTOR of the codes), the layout index.py's comment cites."""
    n_floats = rows.size if n_floats is None else n_floats
    return (fourcc + struct.pack("<i", d) + struct.pack("<q", ntotal) + struct.pack("<q", 1 << 20) * 2 +
            struct.pack("<B", 1) + struct.pack("<i", metric_type) + struct.pack("<Q", n_floats) +
            rows.astype("<f4").tobytes())


def test_read_index_header_against_hand_assembled_faiss_files(tmp_path):
    import io
    rows = np.arange(15, dtype=np.float32).reshape(3, 5)
    for fourcc, mt, name in ((b"IxF2", 1, "l2"), (b"IxFI", 0, "ip")):
        raw = _faiss_flat_bytes(fourcc, 5, 3, mt, rows)
        f = io.BytesIO(raw)
        assert pra.read_index_header(f) == (5, 3, name)
        assert np.frombuffer(f.read(), "<f4").tolist() == rows.ravel().tolist()   # positioned at row 0
    # byte-level golden of the writer: the same bytes, from the same layout
    class Fake:
        d, ntotal, metric = 5, 3, pra.index.METRIC_IP
        def reconstruct_n(self, a, n): return rows[a:a + n]
    p = tmp_path / "w.bin"
    pra.write_index(Fake(), str(p), chunk_rows=2)                         # streamed in 2 chunks
    assert p.read_bytes() == _faiss_flat_bytes(b"IxFI", 5, 3, 0, rows)
    # errors: wrong fourcc, truncated header, header/payload mismatch
    for bad, msg in ((b"IxHN" + raw[4:], "fourcc"), (raw[:20], "truncated"),
                     (_faiss_flat_bytes(b"IxF2", 5, 3, 1, rows, n_floats=14), "holds 14 floats")):
        try:
            pra.read_index_header(io.BytesIO(bad))
        except ValueError as e:
            assert msg in str(e)
        else:
            raise AssertionError(f"accepted a bad header ({msg})")


def test_search_plan_over_the_shape_grid():
    """The dispatch of prag_index_search is ONE pure function (plan_search, flat_index.hip) that runs without a
    GPU: walk the whole shape grid and check the invariants the kernels rely on - tile heights, candidate depth,
    padding, the routes round 4 closed because their kernels spilled, and that bytes / passes are what the
    chosen kernel family reads (bench.py prices its roofline with them)."""
    fams = {"scan8_kernel", "scan_topk_kernel", "scan_qs_kernel", "scan_mm_kernel",
            "scan_mm_kernel<int8 tiles over the 8-bit shadow>", "exact_scan_kernel"}
    n_plans = 0
    for d in (64, 256, 320, 512, 768, 1024, 1536):
        for store in ("f16", "f32"):
            for metric in ("l2", "ip", "cos"):
                for N in (1, 5000, 1 << 20, 2_625_000, 21_000_000):
                    for shadow in (0, 1, 2):
                        prev_ws = 0
                        for B in (1, 32, 33, 64, 65, 128, 129, 1000, 4097):
                            for k in (1, 5, 10, 12, 13, 26, 27, 100, 911):
                                p = pra.plan_search(d, metric, store, N, B, k, shadow)
                                n_plans += 1
                                assert p["family"] in fams, p
                                assert p["QT"] in (32, 64, 128, 256) and p["Bpad"] % p["QT"] == 0 and p["Bpad"] >= B
                                assert p["kc"] >= min(k, 32) and (p["kc"] >= k or p["exact_only"])
                                assert p["grid"] >= 1 and p["launches"] >= 1 and p["bytes_per_launch"] > 0 and p["ws_bytes"] > 0
                                assert not (p["QT"] == 64 and p["kc"] == 32)             # 128 list registers per lane
                                if p["family"] == "scan_qs_kernel":
                                    assert store == "f16" and not (d == 1024 and p["kc"] == 16) and 64 < B <= 128
                                if p["tiled"]:
                                    assert p["QT"] == 256 and d in (256, 512, 768, 1024)
                                if p["int8_tiles"]:
                                    assert shadow and B > 128 and p["kc"] == 256 and k <= 32 and (N >= 2 << 20 or shadow == 2)
                                if p["shadow"]:
                                    assert shadow and d % 128 == 0 and d <= 1024 and B <= 128 and p["family"] == "scan8_kernel"
                                    # HBM-bound tiles (<= 64 queries) are planned on 7/8 of the CUs - which of 7/8 and all
                                    # an index really uses is timed on its own searches and written back into the plan on
                                    # record (test_gpu_shadow.py) -, the epilogue-bound 128-query tiles on every CU
                                    # (round 6: the largest PRIME not above that count - 223 / 251 of 256 - so that rows repeating
                                    #  with a power-of-two period do not all land in two workgroups' candidate regions)
                                    assert p["grid"] <= (223 if p["QT"] <= 64 else 251)
                                    if N >= 1 << 20:
                                        assert p["grid"] == (223 if p["QT"] <= 64 else 251)
                                if p["hp"]:
                                    assert p["QT"] == 32
                                elt = 4 if store == "f32" else 2
                                norms = 4 * N if metric == "l2" else 0
                                want = {"scan8_kernel": N * (d + 12), "scan_topk_kernel": N * d * elt + norms,
                                        "scan_qs_kernel": N * d * 2 + norms, "scan_mm_kernel": N * d * 2 + norms,
                                        "exact_scan_kernel": N * d * elt}.get(p["family"])
                                if want is not None:
                                    assert p["bytes_per_launch"] == want, p
                                if p["family"] in ("scan8_kernel", "scan_topk_kernel"):
                                    assert p["launches"] == p["Bpad"] // p["QT"]
    assert n_plans > 20_000
    # the headline, the reference's call, config 3 and the 8-GPU shard
    assert pra.plan_search(768, "cos", "f16", 21_000_000, 64, 10, 1)["family"] == "scan8_kernel"
    ref = pra.plan_search(768, "l2", "f32", 21_000_000, 1, 5, 0)
    assert (ref["family"], ref["QT"], ref["kc"], ref["hp"], ref["bytes_per_launch"]) == ("scan_topk_kernel", 32, 8, 1, 21_000_000 * 3076)
    c3 = pra.plan_search(768, "cos", "f16", 1_000_000, 1000, 10, 0)
    assert (c3["family"], c3["Bpad"], c3["launches"]) == ("scan_mm_kernel", 1024, 5)      # 2048 -> x16 -> x4 -> x4 -> the rest
    # the segment schedule of the tiled scans is priced where it is decided (bench.py re-derived x16
    # everywhere): 2048, x16, then x4 per segment on the fp16 tiles; x3 everywhere on the int8 tiles
    assert (c3["mm_growth"], c3["last_seg_rows"]) == (16, 1_000_000 - 524_288)
    m3 = pra.plan_search(768, "cos", "f16", 3_000_000, 1000, 10, 0)
    assert (m3["launches"], m3["last_seg_rows"]) == (6, 3_000_000 - 2_097_152)       # not the 2.47 M of a x16 schedule
    m21 = pra.plan_search(768, "cos", "f16", 21_000_000, 1000, 10, 0)
    assert m21["last_seg_rows"] == 21_000_000 - 8_388_608
    i8 = pra.plan_search(768, "cos", "f16", 21_000_000, 1000, 10, 1)
    seg, prev = 2048, 0
    while seg < 21_000_000:
        prev, seg = seg, seg * 3
    assert (i8["int8_tiles"], i8["mm_growth"], i8["last_seg_rows"]) == (1, 3, 21_000_000 - prev)
    assert pra.plan_search(768, "cos", "f16", 1000, 300, 10, 0)["last_seg_rows"] == 1000     # one segment
    assert pra.plan_search(768, "cos", "f16", 2_625_000, 1000, 10, 1)["int8_tiles"] == 1
    assert pra.plan_search(768, "cos", "f16", 1_000_000, 1000, 10, 1)["int8_tiles"] == 0       # below 2 Mi rows
    with __import__("pytest").raises(pra.PragError, match="911"):
        pra.plan_search(768, "cos", "f16", 1000, 1, 912)


def test_plan_lines_carry_what_ran():
    """prag_index_last_plan is a line of key=value fields; execution appends what a plan cannot know (the launch that
    carried a gate) - the parser keeps every field, ints where they parse."""
    from probing_rag_amd.index import parse_plan
    p = parse_plan("family=scan8_kernel QT=64 kc=16 Bpad=64 grid=224 launches=1 bytes_per_launch=16380000000 store=f16 "
                   "metric=2 rows=21000000 queries=64 k=10 gate_launch=scan8_gate_kernel")
    assert p["family"] == "scan8_kernel" and p["grid"] == 224 and p["bytes_per_launch"] == 16380000000
    assert p["gate_launch"] == "scan8_gate_kernel" and p["store"] == "f16"
    p = parse_plan("family=scan_mm_kernel<int8 tiles over the 8-bit shadow> QT=256 kc=256 tiled=1")
    assert p["family"] == "scan_mm_kernel<int8 tiles over the 8-bit shadow>" and p["tiled"] == 1
