"""ctypes loader for the C oracle (test infrastructure; see prag_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

from . import oracle_np as onp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libprag_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "prag_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def prober_forward(state: dict, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    B, d = x.shape
    arrs = [np.ascontiguousarray(state[k], np.float32) for k in onp.STATE_KEYS]
    h = arrs[2].shape[0]
    c = arrs[10].shape[0]
    ptrs = (ctypes.POINTER(ctypes.c_float) * 12)(*[_fp(a) for a in arrs])
    out = np.empty((B, c), np.float32)
    rc = lib().oracle_prober_forward(_fp(x), B, d, h, c, ptrs, _fp(out))
    assert rc == 0
    return out


def gate(logits: np.ndarray, ablation: int = 0, theta: float = 0.0):
    lg = np.ascontiguousarray(logits, np.float32)
    L, B, _ = lg.shape
    ps = np.empty((B, 2), np.float32)
    dec = np.empty((B,), np.int32)
    rc = lib().oracle_gate(_fp(lg), L, B, int(ablation), ctypes.c_double(theta), _fp(ps),
                           dec.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    assert rc == 0
    return ps, dec


def flat_search(xs: np.ndarray, q: np.ndarray, k: int, metric: int = onp.METRIC_L2,
                id_offset: int = 0):
    xs = np.ascontiguousarray(xs, np.float32)
    if metric == onp.METRIC_COS:
        q = onp.normalize_rows(q)
        metric = onp.METRIC_IP
    q = np.ascontiguousarray(q, np.float32)
    N, d = xs.shape
    B = q.shape[0]
    D = np.empty((B, k), np.float32)
    I = np.empty((B, k), np.int64)
    rc = lib().oracle_flat_search(_fp(xs), ctypes.c_int64(N), d, _fp(q), B, k, int(metric),
                                  ctypes.c_int64(id_offset), _fp(D),
                                  I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    assert rc == 0
    return D, I
