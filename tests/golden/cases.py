"""Deterministic case definitions shared by the golden generator and the tests.

Weights / inputs are synthesised from ``oracle_np.synth_rows`` (a pure integer
hash -> float generator) so fixtures need to hold only the reference's outputs.
"""
import numpy as np

from oracle import oracle_np as onp

HIDDEN = 512  # ImprovedProbe default hidden_size, utils.py:30
THETAS = (-2.0, -1.0, 0.0, 1.0, 2.0)   # exp_clf_performance.py:561
ABLATIONS = (0, 1, 2, 3, 4, 5)          # exp_rag.py:409, 574

PROBER_CASES = [
    # C1-shaped and the reference's real call shape (B=1), plus a ragged batch
    dict(name="g2b_B1_s1",    d=2048, L=6, B=1,   sigma=1.0,  wseed=100, xseed=1234),
    dict(name="g2b_B7_s30",   d=2048, L=6, B=7,   sigma=30.0, wseed=100, xseed=1235),
    dict(name="g2b_B128_s1",  d=2048, L=6, B=128, sigma=1.0,  wseed=100, xseed=1236),
    dict(name="g2b_B45_off",  d=2048, L=6, B=45,  sigma=3.0,  wseed=200, xseed=1237, mean=2.5),
    # Mistral-7B width (utils.py:303-304 has a d_model=4096 branch)
    dict(name="m7b_B5_s1",    d=4096, L=2, B=5,   sigma=1.0,  wseed=300, xseed=1238),
]

POOL_CASES = [
    dict(name="pool_B8_T48", d=2048, B=8, T=48, wseed=400, xseed=77),
]


def _u(seed, rows, cols, scale):
    return (onp.synth_rows(seed, 0, rows, cols) * np.float32(scale)).astype(np.float32)


def synth_state(seed: int, d: int) -> dict:
    """A full ImprovedProbe state dict (keys/shapes: SURVEY.md §5 'Checkpoint')
    with non-trivial LayerNorm affines so that folding bugs show."""
    s = seed * 16
    h = HIDDEN
    return {
        "layer_norm_input.weight": (1.0 + _u(s + 0, 1, d, 0.1))[0],
        "layer_norm_input.bias": _u(s + 1, 1, d, 0.1)[0],
        "fc1.weight": _u(s + 2, h, d, 0.6 / np.sqrt(d)),
        "fc1.bias": _u(s + 3, 1, h, 0.02)[0],
        "layer_norm1.weight": (1.0 + _u(s + 4, 1, h, 0.1))[0],
        "layer_norm1.bias": _u(s + 5, 1, h, 0.1)[0],
        "fc2.weight": _u(s + 6, h, h, 0.6 / np.sqrt(h)),
        "fc2.bias": _u(s + 7, 1, h, 0.02)[0],
        "layer_norm2.weight": (1.0 + _u(s + 8, 1, h, 0.1))[0],
        "layer_norm2.bias": _u(s + 9, 1, h, 0.1)[0],
        "fc3.weight": _u(s + 10, 2, h, 0.6 / np.sqrt(h)),
        "fc3.bias": _u(s + 11, 1, 2, 0.02)[0],
    }


def synth_x(seed: int, L: int, B: int, d: int, sigma: float, mean: float = 0.0) -> np.ndarray:
    x = onp.synth_rows(seed, 0, L * B, d).reshape(L, B, d)
    return (x * np.float32(sigma) + np.float32(mean)).astype(np.float32)


def case_x(case: dict) -> np.ndarray:
    return synth_x(case["xseed"], case["L"], case["B"], case["d"], case["sigma"],
                   case.get("mean", 0.0))


def synth_pool_inputs(case: dict):
    B, T, d = case["B"], case["T"], case["d"]
    acts = onp.synth_rows(case["xseed"], 0, B * T, d).reshape(B, T, d).astype(np.float32)
    pred_lens = np.array([1 + (7 * i + 3) % (T - 1) for i in range(B)], dtype=np.int64)
    labels = np.array([i % 2 for i in range(B)], dtype=np.int64)
    return acts, pred_lens, labels


# prober training (utils.py:191-197 method_2_train with train.py:131-135's AdamW / ExponentialLR)
TRAIN_CASES = [
    dict(name="train_B8_T24", d=2048, B=8, T=24, wseed=500, xseed=88, steps=3, seed=4242),
    dict(name="train_B3_T9_nodrop", d=2048, B=3, T=9, wseed=501, xseed=89, steps=2, seed=1, dropout_p=0.0),
]
# the other two training methods of train.py:354 (`each_token` - the script's default -, utils.py:164-173, and
# `last_token`, utils.py:213-220): same optimiser / scheduler, different rows
TRAIN_METHOD_CASES = [
    dict(name="train_each_token_B4_T12", method="each_token", d=2048, B=4, T=12, wseed=502, xseed=90, steps=2, seed=77),
    dict(name="train_last_token_B8_T6", method="last_token", d=2048, B=8, T=6, wseed=503, xseed=91, steps=2, seed=78),
]
TRAIN_SAMPLE_STRIDE = 251   # fc1/fc2 weights are stored as every 251st element (+ their float64 sums)


def synth_train_batch(case: dict, step: int):
    """(acts [B,T,d] f32, pred_lens int64 [B], labels int64 [B]) of optimiser step `step` (1-based)."""
    sub = dict(case, xseed=case["xseed"] + 1000 * step)
    return synth_pool_inputs(sub)


# the retrieval wrappers (utils.py:374-380 find_topk_sim / batch_topk_sim): a stub encoder whose `encode`
# returns preset float32 embeddings (sentence-transformers is not in the image: embeddings are inputs,
# SURVEY.md section 8c) and a recording index that notes what `index.search` is handed
TOPK_CASE = dict(name="topk_N3000_B5", N=3000, d=768, B=5, k=5, xseed=42, qseed=7)


class StubEncoder:
    """`model_retr.encode(query)`: a list of strings -> float32 [len(query), d]; one string -> float32 [d]
    (what SentenceTransformer.encode returns for a str)."""
    def __init__(self, emb):
        self.emb = np.ascontiguousarray(emb, dtype=np.float32)
        self.calls = []

    def encode(self, query):
        self.calls.append(type(query).__name__)
        if isinstance(query, str):
            return self.emb[0].copy()
        return self.emb[: len(query)].copy()


class RecordingIndex:
    """Wraps any object with `search(x, k)`; records how the wrapper called it."""
    def __init__(self, inner):
        self.inner = inner
        self.calls = []

    def search(self, *args, **kwargs):
        x = args[0]
        self.calls.append("args=%d kwargs=%s type=%s dtype=%s shape=%s c_contiguous=%s" % (
            len(args), sorted(kwargs), type(x).__name__, getattr(x, "dtype", None), tuple(getattr(x, "shape", ())),
            bool(getattr(x, "flags", None) is not None and x.flags["C_CONTIGUOUS"])))
        k = kwargs["k"] if "k" in kwargs else args[1]
        return self.inner.search(x, k)


class OracleIndex:
    """The oracle's flat search behind the duck-typed `search(x, k)` of faiss.IndexFlatL2."""
    def __init__(self, xs, metric=onp.METRIC_L2):
        self.xs, self.metric = xs, metric

    def search(self, x, k):
        return onp.flat_search(self.xs, np.asarray(x, dtype=np.float32), k, self.metric)


def topk_inputs(case=TOPK_CASE):
    X = onp.synth_rows(case["xseed"], 0, case["N"], case["d"])
    Q = onp.synth_rows(case["qseed"], 0, case["B"], case["d"])
    Q[1] = X[17] + np.float32(0.01) * Q[1]      # a planted neighbour
    return X, Q
