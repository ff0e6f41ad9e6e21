"""Synthetic inputs for benchmarks and demos, produced by the library itself.

The corpus generator is the counter-based hash of ``prag_index_add_synthetic`` (rows keyed by
(seed, row, column): any shard can generate its own rows, any row can be regenerated anywhere).
``synth_rows`` exposes the same stream as a host array by generating into a scratch index on the
GPU and reading it back, so benchmark drivers need nothing from the test tree.
"""
import numpy as np


def synth_rows(seed: int, row0: int, n: int, d: int) -> np.ndarray:
    """float32 [n,d]: rows [row0,row0+n) of stream `seed` (what ``add_synthetic`` appends)."""
    from .index import HipFlatIndex
    ix = HipFlatIndex(d, "ip", "f32", capacity=max(n, 1))
    ix.add_synthetic(seed, row0, n)
    out = ix.reconstruct_n(0, n)
    ix.close()
    return out


def random_prober_state(seed: int, d_model: int, hidden: int = 512, num_classes: int = 2) -> dict:
    """A randomly initialised ``ImprovedProbe`` state dict (utils.py:29-44: nn.Linear's default
    uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) init; LayerNorm affines perturbed around (1, 0) so that
    the folded-affine paths are exercised).  Checkpoints are not shipped with the reference."""
    import torch
    g = torch.Generator().manual_seed(int(seed))

    def uni(shape, bound):
        return ((torch.rand(shape, generator=g) * 2 - 1) * bound).numpy().astype(np.float32)

    def lin(out_f, in_f):
        b = 1.0 / np.sqrt(in_f)
        return uni((out_f, in_f), b), uni((out_f,), b)

    w1, b1 = lin(hidden, d_model)
    w2, b2 = lin(hidden, hidden)
    w3, b3 = lin(num_classes, hidden)
    return {
        "layer_norm_input.weight": 1.0 + uni((d_model,), 0.1), "layer_norm_input.bias": uni((d_model,), 0.1),
        "fc1.weight": w1, "fc1.bias": b1,
        "layer_norm1.weight": 1.0 + uni((hidden,), 0.1), "layer_norm1.bias": uni((hidden,), 0.1),
        "fc2.weight": w2, "fc2.bias": b2,
        "layer_norm2.weight": 1.0 + uni((hidden,), 0.1), "layer_norm2.bias": uni((hidden,), 0.1),
        "fc3.weight": w3, "fc3.bias": b3,
    }


# ---------------------------------------------------------------------------------------------------------------
# Embedding-shaped rows: what a sentence encoder such as the reference's contriever-msmarco puts out
# un-normalised (make_indexer.py:447-456) does NOT look like iid N(0,1) - the rows share a mean direction, the
# variance around it lives in a few directions, and a handful of fixed coordinates carry values an order of
# magnitude above the rest.  rows = mu + U diag(lambda) z:
#   U       a fixed random rotation (seeded, the same for every shard and for the queries)
#   lambda  power law, lambda_j ~ (j + 1)^-alpha, scaled so that E||U diag(lambda) z|| = resid_norm
#   mu      a dense random direction plus `n_outlier` fixed coordinates at `outlier_ratio` (lo .. hi) x the median
#           |x_j| of the other coordinates (same sign on every row, 10 % relative jitter); ||mu|| = mean_frac of the
#           row norm
# The structure (U, lambda, mu, outlier columns) comes from NumPy on the host (seeded, a few ms); the n x d normals and
# the n x d x d product run wherever `device` says (PyTorch: plumbing).  oracle/oracle_np.py restates the same
# distribution in NumPy for the CPU tests; parity tests feed the oracle the rows the index actually stores.
# ---------------------------------------------------------------------------------------------------------------
def embedding_structure(seed: int, d: int, mean_frac: float = 0.8, alpha: float = 0.5, n_outlier: int = 6,
                        outlier_ratio=(10.0, 30.0), resid_norm: float = 0.6):
    """(U [d,d], lam [d], mu [d], outlier columns) as float32 / int64 NumPy arrays; row norm ~ 1."""
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
    # ||mu||^2 = m_b^2 + sum a_o^2 = (mean_frac * row_norm)^2 with row_norm^2 = ||mu||^2 + resid_norm^2, and
    # a_o = ratio_o * median|x_j|, median|x_j| ~ 0.6745 * sqrt((resid_norm^2 + m_b^2) / d)  ->  solve for m_b
    mu2 = mean_frac ** 2 * resid_norm ** 2 / (1.0 - mean_frac ** 2)
    kappa = (ratios ** 2).sum() * 0.6745 ** 2 / d
    mb2 = (mu2 - kappa * resid_norm ** 2) / (1.0 + kappa)
    if mb2 < 0.0:            # the outlier columns alone exceed mean_frac: no dense part, ||mu|| is what they give
        mb2 = 0.0
    med = 0.6745 * np.sqrt((resid_norm ** 2 + mb2) / d)
    mu = np.sqrt(mb2) * dense
    mu[cols] = signs * ratios * med
    return U.astype(np.float32), lam.astype(np.float32), mu.astype(np.float32), cols.astype(np.int64)


def embedding_like_rows(seed: int, row0: int, n: int, d: int, device="cuda", structure=None, chunk: int = 1 << 18,
                        **kw):
    """float32 [n,d] torch tensor on `device`: rows [row0, row0+n) of the embedding-shaped stream `seed` (any shard
    can generate its own rows: the normals of chunk c are keyed by (seed, c), chunks are aligned to `chunk` rows)."""
    import torch
    U, lam, mu, cols = structure if structure is not None else embedding_structure(seed, d, **kw)
    dev = torch.device(device)
    Ut = torch.from_numpy(np.ascontiguousarray((U * lam[None, :]).T)).to(dev)      # z @ (U diag(lam))^T
    mut = torch.from_numpy(mu).to(dev)
    jit = torch.zeros((d,), device=dev)
    jit[torch.from_numpy(cols).to(dev)] = 0.1
    out = torch.empty((n, d), dtype=torch.float32, device=dev)
    c0 = row0 // chunk
    done = 0
    while done < n:
        c = (row0 + done) // chunk
        lo = c * chunk
        g = torch.Generator(device=dev).manual_seed((int(seed) << 20) + c)
        z = torch.randn((chunk, d + 1), generator=g, device=dev)
        a, b = row0 + done - lo, min(chunk, row0 + n - lo)
        zz = z[a:b]
        rows = zz[:, :d] @ Ut + mut * (1.0 + jit * zz[:, d:])
        out[done:done + (b - a)] = rows
        done += b - a
    del c0
    return out
