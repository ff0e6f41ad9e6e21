"""CPU oracle (NumPy) for the Probing-RAG retrieval-gating hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``probing-rag_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

Every function restates one piece of the reference (paths relative to
``/root/reference``) in float64 arithmetic on float32 (or float16-rounded)
inputs, i.e. "the mathematically exact value, rounded once at the end":

* prober forward ............ utils.py:29-57   (ImprovedProbe)
* gate ...................... exp_rag.py:393, 406-415
* sum / mean pooling ........ exp_rag.py:385-386, train.py:153-162, 202-205
* retrieve-decide loop ...... exp_rag.py:417-468
* flat search ............... utils.py:378-380 -> faiss.IndexFlatL2.search
                              (make_indexer.py:449-457)

Pinning status
--------------
* prober / gate / pooling / loop: pinned against golden vectors produced by
  importing the reference's own ``utils.ImprovedProbe`` and helpers
  (``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``).
* flat search: the arithmetic lives in faiss-cpu (unpinned version,
  ``pip install faiss-cpu`` README.md:23), which is not in the reference tree
  and not installable here.  **parity unpinned**: the oracle follows the
  published definition of IndexFlatL2 / IndexFlatIP (exact brute force,
  squared L2 ascending / inner product descending, 0-based insertion ids,
  -1 padding when ntotal < k) and breaks ties by lowest id.
"""
from __future__ import annotations

import numpy as np

LN_EPS = 1e-5  # torch.nn.LayerNorm default, utils.py:32,43-44

METRIC_L2 = 0
METRIC_IP = 1
METRIC_COS = 2

STATE_KEYS = (
    "layer_norm_input.weight", "layer_norm_input.bias",
    "fc1.weight", "fc1.bias",
    "layer_norm1.weight", "layer_norm1.bias",
    "fc2.weight", "fc2.bias",
    "layer_norm2.weight", "layer_norm2.bias",
    "fc3.weight", "fc3.bias",
)


# --------------------------------------------------------------------------
# prober (utils.py:29-57)
# --------------------------------------------------------------------------
def _layer_norm(x, w, b):
    # torch LayerNorm: biased variance, eps inside the sqrt
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + LN_EPS) * w + b


def _silu(x):
    return x / (1.0 + np.exp(-x))


def prober_forward(state: dict, x: np.ndarray) -> np.ndarray:
    """ImprovedProbe.forward in eval mode (dropout = identity, utils.py:329).

    Order is Linear -> SiLU -> LayerNorm (post-activation norm), utils.py:48-56.
    x: [B, d_model]; returns float32 logits [B, n_classes].
    """
    s = {k: np.asarray(v, dtype=np.float64) for k, v in state.items()}
    h = np.asarray(x, dtype=np.float64)
    h = _layer_norm(h, s["layer_norm_input.weight"], s["layer_norm_input.bias"])
    h = h @ s["fc1.weight"].T + s["fc1.bias"]
    h = _layer_norm(_silu(h), s["layer_norm1.weight"], s["layer_norm1.bias"])
    h = h @ s["fc2.weight"].T + s["fc2.bias"]
    h = _layer_norm(_silu(h), s["layer_norm2.weight"], s["layer_norm2.bias"])
    out = h @ s["fc3.weight"].T + s["fc3.bias"]
    return out.astype(np.float32)


def ensemble_forward(states: list, x: np.ndarray) -> np.ndarray:
    """x: [L, B, d]; one independent weight set per probed layer
    (utils.py:385-390).  Returns float32 [L, B, 2]."""
    return np.stack([prober_forward(s, x[l]) for l, s in enumerate(states)])


# --------------------------------------------------------------------------
# gate (exp_rag.py:393, 406-415)
# --------------------------------------------------------------------------
def gate(logits: np.ndarray, ablation: int = 0, theta: float = 0.0):
    """logits [L, B, 2] -> (probsum float32 [B, 2], decision int32 [B]).

    ``for num in range(args.ablation, len(logits)): s += softmax(logits[num])``
    then ``0 if s[0].item() + theta < s[1].item() else 1`` (1 = retrieve,
    exp_rag.py:414).  The reference sums in float32 on CPU, layer by layer in
    increasing order, and compares Python floats (the float32 sums widened to
    double, theta a double); the restatement keeps both so that decisions are
    comparable bit for bit.
    """
    lg = np.asarray(logits, dtype=np.float32)
    L, B, _ = lg.shape
    s = np.zeros((B, 2), dtype=np.float32)
    for n in range(ablation, L):
        z = lg[n].astype(np.float64)
        z = z - z.max(axis=1, keepdims=True)
        e = np.exp(z)
        p = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
        s = (s + p).astype(np.float32)
    decision = np.where((s[:, 0].astype(np.float64) + float(theta)) < s[:, 1].astype(np.float64), 0, 1).astype(np.int32)
    return s, decision


# --------------------------------------------------------------------------
# pooling (exp_rag.py:385-386 sum ; train.py:153-162,202-205 ragged mean)
# --------------------------------------------------------------------------
def pool_sum_decode_steps(cache_list: list) -> np.ndarray:
    """exp_rag.py:385-386: cat(cache[name][1:], dim=1).sum(dim=1).
    cache_list[0] is the prompt pass [1,P,d] and is skipped; raises like
    torch.concat([]) does when only one forward pass happened."""
    if len(cache_list) < 2:
        raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")
    cat = np.concatenate([np.asarray(c, dtype=np.float64) for c in cache_list[1:]], axis=1)
    return cat.sum(axis=1).astype(np.float32)


def pool_ragged_mean(acts: np.ndarray, pred_lens: np.ndarray) -> np.ndarray:
    """train.py:153-162 + 202-205: mean of the last pred_len[i] positions of
    acts[i] ([B,T,d]) -> [B,d]."""
    a = np.asarray(acts, dtype=np.float64)
    out = np.empty((a.shape[0], a.shape[2]), dtype=np.float64)
    for i, n in enumerate(np.asarray(pred_lens).tolist()):
        out[i] = a[i, a.shape[1] - n:, :].mean(axis=0)
    return out.astype(np.float32)


# --------------------------------------------------------------------------
# retrieve-decide loop control flow (exp_rag.py:417-468)
# --------------------------------------------------------------------------
def retr_count_from_decisions(decisions: list) -> tuple:
    """decisions[0] is the gate after the first (no-retrieval) generation,
    decisions[j>0] the gate after retrieval round j.  Returns
    (retr_count, rounds_run) as the reference records them: at most 4 rounds,
    retr_count saturates at 3 (exp_rag.py:462-465)."""
    if decisions[0] == 0:
        return 0, 0
    retr_count = 0
    rounds = 0
    d = 1
    while d == 1:
        rounds += 1
        d = decisions[rounds] if rounds < len(decisions) else 1
        if retr_count > 2:
            break
        retr_count += 1
    return retr_count, rounds


# --------------------------------------------------------------------------
# flat index (faiss.IndexFlatL2 / IP definition; utils.py:378-380)
# --------------------------------------------------------------------------
def normalize_rows(x: np.ndarray) -> np.ndarray:
    """COSINE mode: rows are L2-normalised in float64 and rounded to float32."""
    x64 = np.asarray(x, dtype=np.float64)
    n = np.sqrt((x64 * x64).sum(axis=1, keepdims=True))
    n = np.where(n > 0, n, 1.0)
    return (x64 / n).astype(np.float32)


def store_round(x: np.ndarray, store_dtype: str) -> np.ndarray:
    """What the index keeps: float32 as given, or round-to-nearest-even fp16."""
    x = np.asarray(x, dtype=np.float32)
    if store_dtype == "f16":
        return x.astype(np.float16).astype(np.float32)
    return x


def flat_scores(xs: np.ndarray, q: np.ndarray, metric: int) -> np.ndarray:
    """Exact scores in float64: xs [N,d] are the STORED rows (already
    normalised / rounded), q [B,d] float32.  L2 -> squared distance."""
    x64 = xs.astype(np.float64)
    q64 = np.asarray(q, dtype=np.float64)
    if metric == METRIC_COS:
        q64 = normalize_rows(q).astype(np.float64)
    if metric == METRIC_L2:
        # sum_c (q_c - x_c)^2, evaluated without the norm expansion
        out = np.empty((q64.shape[0], x64.shape[0]), dtype=np.float64)
        for b in range(q64.shape[0]):
            diff = x64 - q64[b]
            out[b] = np.einsum("nd,nd->n", diff, diff)
        return out
    return q64 @ x64.T


def flat_search(xs: np.ndarray, q: np.ndarray, k: int, metric: int = METRIC_L2,
                id_offset: int = 0):
    """index.search(q, k) -> (D float32 [B,k], I int64 [B,k]).

    L2: k smallest squared distances ascending; IP/COS: k largest descending.
    Ties -> lowest id first (stable sort).  Fewer than k rows -> D padded with
    +/-FLT_MAX-like inf sentinel and I with -1, as faiss does.
    """
    sc = flat_scores(xs, q, metric)
    B, N = sc.shape
    key = sc if metric == METRIC_L2 else -sc
    if metric == METRIC_L2:
        order = np.argsort(key, axis=1, kind="stable")[:, :k]
    else:
        # flat_scores uses a BLAS matmul for inner products, whose rounding depends on where a
        # row sits in the blocked GEMM: exact duplicates can differ in the last bit and would
        # then be ordered by that noise instead of by id.  Preselect generously with the BLAS
        # scores, rescore the preselection row by row (same summation order for every row, so
        # identical rows get identical scores) and order by (score, id).
        m = min(N, k + 64)
        pre = np.argsort(key, axis=1, kind="stable")[:, :m]
        x64 = xs.astype(np.float64)
        q64 = (normalize_rows(q) if metric == METRIC_COS else np.asarray(q)).astype(np.float64)
        order = np.empty((B, min(N, k)), dtype=np.int64)
        for b in range(B):
            ids = np.sort(pre[b])
            exact = np.einsum("nd,d->n", x64[ids], q64[b])
            sc[b, ids] = exact
            order[b] = ids[np.argsort(-exact, kind="stable")[:k]]
    D = np.take_along_axis(sc, order, axis=1).astype(np.float32)
    I = order.astype(np.int64) + id_offset
    if N < k:
        pad = k - N
        fill = np.float32(np.finfo(np.float32).max if metric == METRIC_L2
                          else -np.finfo(np.float32).max)
        D = np.concatenate([D, np.full((B, pad), fill, np.float32)], axis=1)
        I = np.concatenate([I, np.full((B, pad), -1, np.int64)], axis=1)
    return D, I


def merge_topk(D_parts: list, I_parts: list, k: int, metric: int):
    """k-way merge of per-shard (D, I) by (score, id) — the exchange step of
    the row-sharded index.  Padding entries (I == -1) sort last."""
    D = np.concatenate(D_parts, axis=1)
    I = np.concatenate(I_parts, axis=1)
    key = D.astype(np.float64) if metric == METRIC_L2 else -D.astype(np.float64)
    key = np.where(I < 0, np.inf, key)
    idkey = np.where(I < 0, np.iinfo(np.int64).max, I)
    out_D = np.empty((D.shape[0], k), np.float32)
    out_I = np.empty((D.shape[0], k), np.int64)
    for b in range(D.shape[0]):
        o = np.lexsort((idkey[b], key[b]))[:k]
        out_D[b] = D[b, o]
        out_I[b] = I[b, o]
    return out_D, out_I


# --------------------------------------------------------------------------
# synthetic data shared by tests / bench / GPU generator
# --------------------------------------------------------------------------
def _mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32 integer hash on uint32 arrays."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def synth_rows(seed: int, row0: int, nrows: int, d: int) -> np.ndarray:
    """Counter-based N(0,1)-ish rows keyed by (seed,row,col): sum of four
    uniform 16-bit fields (Irwin-Hall, variance 1/3) scaled to unit variance.
    Pure integer -> float arithmetic so the HIP generator reproduces it bit
    for bit and any shard can generate its own rows."""
    rows = (np.arange(row0, row0 + nrows, dtype=np.uint64)[:, None])
    cols = np.arange(d, dtype=np.uint64)[None, :]
    ctr = (rows * np.uint64(d) + cols)
    lo = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (ctr >> np.uint64(32)).astype(np.uint32)
    s = np.uint32(seed & 0xFFFFFFFF)
    h1 = _mix32(lo ^ _mix32(hi ^ s))
    h2 = _mix32(h1 ^ np.uint32(0x9E3779B9))
    a = (h1 & np.uint32(0xFFFF)).astype(np.int32)
    b = (h1 >> np.uint32(16)).astype(np.int32)
    c = (h2 & np.uint32(0xFFFF)).astype(np.int32)
    e = (h2 >> np.uint32(16)).astype(np.int32)
    t = (a + b + c + e - 131070).astype(np.float32)  # exact in f32 (|t| < 2^18)
    # var(sum of 4 U{0..65535}) = 4*(65536^2-1)/12 ; scale to unit variance
    return (t * np.float32(1.0 / 37837.22)).astype(np.float32)


# --------------------------------------------------------------------------
# Embedding-shaped synthetic corpus: the reference indexes UN-NORMALISED contriever output under
# L2 (make_indexer.py:447-456) - rows that share a mean direction, with a power-law spectrum around it and a few
# outlier coordinates.  NumPy restatement of probing_rag_amd/synth.py (same structure; the normals come from
# NumPy's generator here, so the two streams agree in distribution, not bit for bit - parity tests feed the
# oracle the rows the index actually stores).
# --------------------------------------------------------------------------
def embedding_structure(seed: int, d: int, mean_frac: float = 0.8, alpha: float = 0.5, n_outlier: int = 6,
                        outlier_ratio=(10.0, 30.0), resid_norm: float = 0.6):
    rng = np.random.default_rng(int(seed) * 7919 + 13)
    U, _ = np.linalg.qr(rng.standard_normal((d, d)))
    lam = (np.arange(1, d + 1, dtype=np.float64)) ** (-float(alpha))
    lam *= resid_norm / np.sqrt((lam ** 2).sum())
    cols = np.sort(rng.choice(d, size=n_outlier, replace=False)) if n_outlier else np.zeros((0,), np.int64)
    signs = rng.choice([-1.0, 1.0], size=n_outlier)
    ratios = np.linspace(outlier_ratio[0], outlier_ratio[1], n_outlier) if n_outlier else np.zeros((0,))
    dense = rng.standard_normal(d)
    dense[cols] = 0.0
    dense /= np.linalg.norm(dense)
    mu2 = mean_frac ** 2 * resid_norm ** 2 / (1.0 - mean_frac ** 2)
    kappa = (ratios ** 2).sum() * 0.6745 ** 2 / d
    mb2 = max(0.0, (mu2 - kappa * resid_norm ** 2) / (1.0 + kappa))
    med = 0.6745 * np.sqrt((resid_norm ** 2 + mb2) / d)
    mu = np.sqrt(mb2) * dense
    mu[cols] = signs * ratios * med
    return U.astype(np.float32), lam.astype(np.float32), mu.astype(np.float32), cols.astype(np.int64)


def embedding_like_rows(seed: int, row0: int, n: int, d: int, structure=None, chunk: int = 1 << 18, **kw) -> np.ndarray:
    """float32 [n,d]: rows = mu (1 + 0.1 z' on the outlier coordinates) + U diag(lambda) z."""
    U, lam, mu, cols = structure if structure is not None else embedding_structure(seed, d, **kw)
    M = np.ascontiguousarray((U * lam[None, :]).T)
    jit = np.zeros((d,), np.float32)
    jit[cols] = 0.1
    out = np.empty((n, d), np.float32)
    done = 0
    while done < n:
        c = (row0 + done) // chunk
        lo = c * chunk
        z = np.random.default_rng((int(seed) << 20) + c).standard_normal((chunk, d + 1), dtype=np.float32)
        a, b = row0 + done - lo, min(chunk, row0 + n - lo)
        zz = z[a:b]
        out[done:done + (b - a)] = zz[:, :d] @ M + mu[None, :] * (1.0 + jit[None, :] * zz[:, d:])
        done += b - a
    return out


# --------------------------------------------------------------------------
# train/eval-time forward (train.py:141-151, 199-208, 170-181; utils.py:122-189)
# --------------------------------------------------------------------------
def train_eval_forward(state: dict, acts: np.ndarray, pred_lens: np.ndarray,
                       labels: np.ndarray):
    """_method_2_util + return_acc: ragged mean pool -> prober -> softmax(-1)
    -> CrossEntropyLoss applied to the PROBABILITIES (the reference's "double
    softmax", train.py:149-150) -> argmax accuracy.
    Returns (probs float32 [B,2], loss float32, acc float)."""
    pooled = pool_ragged_mean(acts, pred_lens)
    z = prober_forward(state, pooled).astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    p = np.exp(z)
    p = p / p.sum(axis=1, keepdims=True)
    # CrossEntropyLoss(input=p) = mean_i( logsumexp(p_i) - p_i[label_i] )
    lse = np.log(np.exp(p).sum(axis=1))
    lab = np.asarray(labels).astype(np.int64)
    loss = float(np.mean(lse - p[np.arange(len(lab)), lab]))
    acc = float((p.argmax(axis=1) == lab).sum()) / len(lab)
    return p.astype(np.float32), np.float32(loss), acc


def pool_each_token(acts: np.ndarray, pred_lens: np.ndarray, labels: np.ndarray = None):
    """`_input_tensor_method1` (train.py:153-162, utils.py:134-143; the `each_token` method): the last
    pred_lens[i] positions of every sample concatenated in sample order, and
    `torch.repeat_interleave(labels, pred_lens)`."""
    T = acts.shape[1]
    rows = np.concatenate([acts[i, T - int(n):, :] for i, n in enumerate(pred_lens)], axis=0)
    new_labels = None if labels is None else np.repeat(np.asarray(labels), np.asarray(pred_lens).astype(np.int64))
    return rows.astype(np.float32), new_labels


def pool_last_token(acts: np.ndarray) -> np.ndarray:
    """`activations[:, -1, :]` (train.py:228, utils.py:206; the `last_token` method)."""
    return np.ascontiguousarray(acts[:, -1, :], dtype=np.float32)


def eval_forward_rows(state: dict, x: np.ndarray, labels: np.ndarray):
    """make_loss + return_acc (train.py:141-151, 170-181) on already-pooled rows: the tail of
    train_eval_forward, shared by method_1_eval / method_3_eval (utils.py:175-179, 222-226)."""
    z = prober_forward(state, x).astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    p = np.exp(z)
    p = p / p.sum(axis=1, keepdims=True)
    lse = np.log(np.exp(p).sum(axis=1))
    lab = np.asarray(labels).astype(np.int64)
    loss = float(np.mean(lse - p[np.arange(len(lab)), lab]))
    acc = float((p.argmax(axis=1) == lab).sum()) / len(lab)
    return p.astype(np.float32), np.float32(loss), acc


# --------------------------------------------------------------------------
# prober training step (train.py:141-151, 210-220 / utils.py:191-197:
# method_2_train = forward in train mode -> CrossEntropyLoss on the softmax
# PROBABILITIES -> backward -> AdamW.step -> ExponentialLR.step)
# --------------------------------------------------------------------------
ADAMW_DEFAULTS = dict(lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01,  # torch.optim.AdamW defaults,
                      gamma=0.995, dropout_p=0.1)                                     # train.py:131-135, utils.py:39


def dropout_keep(seed: int, step: int, site: int, B: int, n: int, p: float) -> np.ndarray:
    """Bernoulli(1-p) keep mask [B,n] of dropout site `site` (0 after LN1, 1 after LN2) at optimiser
    step `step`, from the counter hash the HIP trainer uses: the reference draws its masks from
    torch's global RNG stream, which cannot be reproduced outside torch, so parity is defined on
    identical masks (the golden generator feeds these masks to the reference's own module)."""
    idx = (np.arange(B, dtype=np.uint64)[:, None] * np.uint64(n) + np.arange(n, dtype=np.uint64)[None, :])
    key = _mix32(np.array([(seed & 0xFFFFFFFF) ^ ((step * 0x9E3779B9) & 0xFFFFFFFF)], dtype=np.uint32))
    site_key = _mix32(key ^ np.uint32(site + 1))[0]
    h = _mix32(idx.astype(np.uint32) ^ site_key)
    thresh = np.uint32(min(int(round(p * 4294967296.0)), 4294967295))
    return h >= thresh


def _ln_fwd(x, g, b):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    r = 1.0 / np.sqrt(var + LN_EPS)
    xh = (x - mu) * r
    return xh * g + b, xh, r


def _ln_bwd(dy, xh, r, g):
    dxh = dy * g
    dx = r * (dxh - dxh.mean(axis=-1, keepdims=True) - xh * (dxh * xh).mean(axis=-1, keepdims=True))
    return dx, (dy * xh).sum(axis=0), dy.sum(axis=0)


def train_loss_and_grads(state: dict, x: np.ndarray, labels: np.ndarray, keep1=None, keep2=None,
                         dropout_p: float = 0.0):
    """Forward in train mode + backward, float64.  keep1/keep2: boolean keep masks [B,512] of the two
    dropout sites (None = no dropout).  Returns (loss, probs [B,C], grads dict keyed like the state)."""
    s = {k: np.asarray(v, dtype=np.float64) for k, v in state.items()}
    x = np.asarray(x, dtype=np.float64)
    B = x.shape[0]
    scale = np.float64(np.float32(1.0 / (1.0 - dropout_p))) if dropout_p > 0 else 1.0
    m1 = (keep1.astype(np.float64) * scale) if keep1 is not None else 1.0
    m2 = (keep2.astype(np.float64) * scale) if keep2 is not None else 1.0
    y0, xh0, r0 = _ln_fwd(x, s["layer_norm_input.weight"], s["layer_norm_input.bias"])
    h1 = y0 @ s["fc1.weight"].T + s["fc1.bias"]
    s1 = _silu(h1)
    n1, sh1, r1 = _ln_fwd(s1, s["layer_norm1.weight"], s["layer_norm1.bias"])
    d1 = n1 * m1
    h2 = d1 @ s["fc2.weight"].T + s["fc2.bias"]
    s2 = _silu(h2)
    n2, sh2, r2 = _ln_fwd(s2, s["layer_norm2.weight"], s["layer_norm2.bias"])
    d2 = n2 * m2
    z = d2 @ s["fc3.weight"].T + s["fc3.bias"]
    z = z - z.max(axis=1, keepdims=True)
    p = np.exp(z)
    p = p / p.sum(axis=1, keepdims=True)
    lab = np.asarray(labels).astype(np.int64)
    ep = np.exp(p)
    sm = ep / ep.sum(axis=1, keepdims=True)                    # softmax of the probabilities (double softmax)
    loss = float(np.mean(np.log(ep.sum(axis=1)) - p[np.arange(B), lab]))
    gp = sm.copy()
    gp[np.arange(B), lab] -= 1.0
    gp /= B
    dz = p * (gp - (gp * p).sum(axis=1, keepdims=True))
    g = {}
    g["fc3.weight"] = dz.T @ d2
    g["fc3.bias"] = dz.sum(axis=0)
    dn2 = (dz @ s["fc3.weight"]) * m2
    ds2, g["layer_norm2.weight"], g["layer_norm2.bias"] = _ln_bwd(dn2, sh2, r2, s["layer_norm2.weight"])
    sg2 = 1.0 / (1.0 + np.exp(-h2))
    dh2 = ds2 * sg2 * (1.0 + h2 * (1.0 - sg2))
    g["fc2.weight"] = dh2.T @ d1
    g["fc2.bias"] = dh2.sum(axis=0)
    dn1 = (dh2 @ s["fc2.weight"]) * m1
    ds1, g["layer_norm1.weight"], g["layer_norm1.bias"] = _ln_bwd(dn1, sh1, r1, s["layer_norm1.weight"])
    sg1 = 1.0 / (1.0 + np.exp(-h1))
    dh1 = ds1 * sg1 * (1.0 + h1 * (1.0 - sg1))
    g["fc1.weight"] = dh1.T @ y0
    g["fc1.bias"] = dh1.sum(axis=0)
    dy0 = dh1 @ s["fc1.weight"]
    g["layer_norm_input.weight"] = (dy0 * xh0).sum(axis=0)
    g["layer_norm_input.bias"] = dy0.sum(axis=0)
    return loss, p, g


def adamw_update(state: dict, grads: dict, opt: dict, step: int, hp: dict):
    """torch.optim.AdamW.step (decoupled weight decay, bias-corrected moments) with the learning rate
    ExponentialLR has reached after `step - 1` scheduler steps; `step` counts from 1.  In place on
    float64 copies held in `state` / `opt` (opt[k] = (exp_avg, exp_avg_sq))."""
    lr = hp["lr"] * hp["gamma"] ** (step - 1)
    b1, b2 = hp["beta1"], hp["beta2"]
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for k in STATE_KEYS:
        p = state[k]
        g = grads[k]
        m, v = opt[k]
        p *= 1.0 - lr * hp["weight_decay"]
        m *= b1
        m += (1.0 - b1) * g
        v *= b2
        v += (1.0 - b2) * g * g
        p -= (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + hp["eps"])
    return lr


def train_steps(state: dict, batches: list, seed: int, hp: dict = None):
    """Run len(batches) optimiser steps (each batch = (x [B,d], labels [B])); returns
    (final state float64, [loss per step], [lr used per step])."""
    hp = dict(ADAMW_DEFAULTS, **(hp or {}))
    st = {k: np.array(v, dtype=np.float64) for k, v in state.items()}
    opt = {k: (np.zeros_like(st[k]), np.zeros_like(st[k])) for k in STATE_KEYS}
    losses, lrs = [], []
    H = st["fc1.bias"].shape[0]
    for t, (x, labels) in enumerate(batches, start=1):
        B = x.shape[0]
        k1 = dropout_keep(seed, t, 0, B, H, hp["dropout_p"]) if hp["dropout_p"] > 0 else None
        k2 = dropout_keep(seed, t, 1, B, H, hp["dropout_p"]) if hp["dropout_p"] > 0 else None
        loss, _, g = train_loss_and_grads(st, x, labels, k1, k2, hp["dropout_p"])
        lrs.append(adamw_update(st, g, opt, t, hp))
        losses.append(loss)
    return st, losses, lrs
