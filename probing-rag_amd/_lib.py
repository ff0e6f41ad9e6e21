"""ctypes binding of libprag.so (include/prag.h).

There is NO fallback: if the HIP library cannot be loaded every entry point
raises.  The reference path being replaced is pure Python + faiss-cpu; this
module is the only way the product computes anything.
"""
import ctypes
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# PRAG_LIB: load another build of the library (the timing-only `make diag` build of the tools)
LIB_PATH = os.environ.get("PRAG_LIB") or os.path.join(_HERE, "lib", "libprag.so")

PRAG_OK = 0
PRAG_F32, PRAG_F16, PRAG_BF16 = 0, 1, 2
PRAG_W_F32, PRAG_W_F16 = 0, 1
METRIC_L2, METRIC_IP, METRIC_COS = 0, 1, 2
_METRICS = {"l2": METRIC_L2, "ip": METRIC_IP, "cos": METRIC_COS, "cosine": METRIC_COS}
_ERR = {-1: "PRAG_EINVAL", -2: "PRAG_EHIP", -3: "PRAG_ENOMEM", -4: "PRAG_EUNSUPPORTED", -5: "PRAG_ESTATE"}


class PragError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_ERR.get(code, code)}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile libprag.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(os.path.dirname(_HERE), "include", "prag.h")]
    stale = (not os.path.exists(LIB_PATH)) or \
        os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if force or stale:
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            raise RuntimeError("libprag.so is missing/stale and hipcc is not available to build it")
        subprocess.check_call(["make", "-s", "-j3", "-C", CSRC])
    return LIB_PATH


_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float
_FP = ctypes.POINTER(ctypes.c_float)

# name -> (restype, argtypes); every symbol declared in include/prag.h
SIGNATURES = {
    "prag_version": (_I, []),
    "prag_last_error": (ctypes.c_char_p, []),
    "prag_prober_create": (_I, [ctypes.POINTER(_P), _I, _I, _I, _I, _I]),
    "prag_prober_load_layer": (_I, [_P, _I] + [_FP] * 12),
    "prag_prober_forward": (_I, [_P, _P, _I, _L, _I, _I, _I, _P, _P]),
    "prag_gate": (_I, [_P, _P, _I, _L, _I, _I, ctypes.c_double, _P, _P, _P, _P]),
    "prag_gate_decide": (_I, [_P, _P, _I, _L, _I, _I, ctypes.c_double, _P, _P, _P]),
    "prag_gate_from_logits": (_I, [_P, _I, _I, _I, ctypes.c_double, _P, _P, _P]),
    "prag_prober_effective_weights": (_I, [_P, _I] + [_FP] * 6),
    "prag_prober_reserve": (_I, [_P, _I]),
    "prag_prober_profile": (_I, [_P, _I]),
    "prag_prober_profile_read": (_I, [_P, _FP, _I, ctypes.POINTER(_I)]),
    "prag_prober_destroy": (None, [_P]),
    "prag_pool_accumulate": (_I, [_P, _P, _I, _L, _I, _P]),
    "prag_pool_accumulate_layers": (_I, [_P, _P, _I, _I, _L, _I, _P]),
    "prag_pool_step_gate": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, ctypes.c_double, ctypes.POINTER(ctypes.c_uint64), _P]),
    "prag_gate_step_result": (_I, [_P, ctypes.c_uint64, _I, _P, _P, _P]),
    "prag_pool_ragged": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "prag_pool_masked_mean": (_I, [_P, _I, _P, _I, _I, _I, _P, _P]),
    "prag_pool_each_token": (_I, [_P, _I, _I, _I, _I, _P, _L, _P, _P, _P, _P]),
    "prag_trainer_create": (_I, [ctypes.POINTER(_P), _I, _I, _I] + [ctypes.c_double] * 7 + [ctypes.c_uint32]),
    "prag_trainer_load": (_I, [_P] + [_FP] * 12),
    "prag_trainer_step": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "prag_trainer_export": (_I, [_P] + [_FP] * 12),
    "prag_trainer_set_training": (_I, [_P, _I]),
    "prag_trainer_lr": (ctypes.c_double, [_P]),
    "prag_trainer_steps": (_L, [_P]),
    "prag_trainer_destroy": (None, [_P]),
    "prag_index_create": (_I, [ctypes.POINTER(_P), _I, _I, _I, _L]),
    "prag_index_add": (_I, [_P, _P, _L, _I, _P]),
    "prag_index_add_synthetic": (_I, [_P, ctypes.c_uint32, _L, _L]),
    "prag_index_ntotal": (_L, [_P]),
    "prag_index_d": (_I, [_P]),
    "prag_index_search": (_I, [_P, _P, _I, _I, _L, _P, _P, _I, _P]),
    "prag_merge_topk": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "prag_merge_topk_packed": (_I, [_P, _L, _I, _I, _I, _I, _P, _P, _P]),
    "prag_merge_topk_packed_tagged": (_I, [_P, _L, _I, _I, _I, _I, _P, _P, _P]),
    "prag_index_search_tagged": (_I, [_P, _P, _I, _I, _L, _P, _P, _I, _P]),
    "prag_index_reconstruct": (_I, [_P, _L, _L, _P]),
    "prag_index_set_candidate_depth": (_I, [_P, _I]),
    "prag_rccl_unique_id": (_I, [_P]),
    "prag_rccl_comm_init_rank": (_I, [ctypes.POINTER(_P), _I, _I, _P]),
    "prag_rccl_all_gather": (_I, [_P, _P, _P, ctypes.c_size_t, _P]),
    "prag_rccl_comm_destroy": (_I, [_P]),
    "prag_index_set_comm": (_I, [_P, _P, _I, _I]),
    "prag_index_search_sharded": (_I, [_P, _P, _I, _I, _L, _P, _P, _P]),
    "prag_plan_search": (_I, [_I, _I, _I, _L, _I, _I, _I, _I, ctypes.c_char_p, _I]),
    "prag_index_last_plan": (_I, [_P, ctypes.c_char_p, _I]),
    "prag_index_last_fallbacks": (_I, [_P, _P, ctypes.POINTER(_I)]),
    "prag_search_and_gate": (_I, [_P, _P, _I, _I, _L, _P, _P, _P, _P, _I, _L, _I, _I, ctypes.c_double, _P, _P, _P, _I, _P]),
    "prag_index_stream_wait_scan": (_I, [_P, _P]),
    "prag_index_last_survivors": (_I, [_P, _P, ctypes.POINTER(_L), ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "prag_index_last_tiled8": (_I, [_P, ctypes.POINTER(_I)]),
    "prag_index_set_shadow": (_I, [_P, _I]),
    "prag_index_prepare": (_I, [_P, _P]),
    "prag_index_reserve": (_I, [_P, ctypes.c_int, ctypes.c_int, _P]),
    "prag_index_set_scan_workgroups": (_I, [_P, _I]),
    "prag_index_set_adaptive": (_I, [_P, _I]),
    "prag_index_profile": (_I, [_P, _I]),
    "prag_index_profile_read": (_I, [_P, _FP, _I, ctypes.POINTER(_I)]),
    "prag_index_profile_read_exchange": (_I, [_P, _FP, _I, ctypes.POINTER(_I)]),
    "prag_index_destroy": (None, [_P]),
}

_lib = None


def lib():
    """Load libprag.so (building it first if it is missing and hipcc exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # fail loudly: there is no CPU fallback
            raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != PRAG_OK:
        raise PragError(rc, lib().prag_last_error().decode("utf-8", "replace"))


def metric_id(metric) -> int:
    if isinstance(metric, str):
        return _METRICS[metric.lower()]
    return int(metric)


def current_stream_ptr(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NoCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_CTX = _NoCtx()


def on_device(device):
    """`with on_device(dev):` = `with torch.cuda.device(dev):` when dev is not the current device, nothing otherwise
    (the context manager costs ~5 us per call: a third of a one-query gate decision)."""
    import torch
    idx = device.index if hasattr(device, "index") else int(device)
    if idx is None or idx == torch.cuda.current_device():
        return _NO_CTX
    return torch.cuda.device(device)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("probing_rag_amd needs an AMD GPU (gfx950): no HIP device is visible and "
                           "there is no CPU fallback for the hot path")
