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
