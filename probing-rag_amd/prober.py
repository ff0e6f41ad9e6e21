"""Host-side mirror of the reference's prober interface on top of libprag.so.

Reference call sites (paths relative to /root/reference):
  * ``ImprovedProbe(input_size, output_size)``; ``.to(device)``,
    ``.load_state_dict(sd)``, ``.eval()``, ``prober(x)``  utils.py:29-57, 302-329
  * ``Config_Maker``                                       utils.py:282-290
  * ``load_prober_cfg_gemma_2b``                           utils.py:382-383
  * ``load_prober_models``                                 utils.py:385-387
  * ``return_prober_logit_gemma_2b``                       utils.py:389-390
  * gate (softmax / sum / threshold)                       exp_rag.py:393, 406-415

``HipProber`` is one layer's prober (drop-in for ``ImprovedProbe`` in eval
mode); ``HipProberEnsemble`` owns all probed layers and runs them plus the gate
in one fused launch — ``ensemble.probers`` is a list of per-layer callables so
the reference's ``zip(cfg_list, probers)`` loop works unchanged.
"""
import ctypes

import numpy as np

from . import _lib

STATE_KEYS = (
    "layer_norm_input.weight", "layer_norm_input.bias",
    "fc1.weight", "fc1.bias",
    "layer_norm1.weight", "layer_norm1.bias",
    "fc2.weight", "fc2.bias",
    "layer_norm2.weight", "layer_norm2.bias",
    "fc3.weight", "fc3.bias",
)
HIDDEN = 512
_WEIGHT_MODES = {"f32": _lib.PRAG_W_F32, "fp32": _lib.PRAG_W_F32, "f16": _lib.PRAG_W_F16, "fp16": _lib.PRAG_W_F16}


def _as_f32_host(v):
    if hasattr(v, "detach"):
        v = v.detach().to("cpu").float().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


def _x_dtype(x):
    import torch
    if x.dtype == torch.float32:
        return _lib.PRAG_F32
    if x.dtype == torch.float16:
        return _lib.PRAG_F16
    if x.dtype == torch.bfloat16:
        return _lib.PRAG_BF16
    raise TypeError(f"activations must be float32, float16 or bfloat16, got {x.dtype}")


class HipProberEnsemble:
    """All probed layers' ImprovedProbe weights + the gate, on one GPU."""

    def __init__(self, n_layers: int, d_model: int, num_classes: int = 2, weights: str = "f32",
                 device=None):
        _lib.require_gpu()
        import torch
        self.n_layers, self.d_model, self.num_classes = int(n_layers), int(d_model), int(num_classes)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.weights = weights
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_prober_create(ctypes.byref(h), self.n_layers, self.d_model, HIDDEN,
                                                     self.num_classes, _WEIGHT_MODES[weights]))
        self._h = h
        self.probers = [_LayerView(self, l) for l in range(self.n_layers)]

    # -- weights ------------------------------------------------------------
    def load_layer(self, layer_idx: int, state_dict: dict):
        missing = [k for k in STATE_KEYS if k not in state_dict]
        unexpected = [k for k in state_dict if k not in STATE_KEYS]
        if missing or unexpected:  # same contract as nn.Module.load_state_dict(strict=True)
            raise RuntimeError(f"Error(s) in loading state_dict for ImprovedProbe: missing {missing}, "
                               f"unexpected {unexpected}")
        arrs = [_as_f32_host(state_dict[k]) for k in STATE_KEYS]
        want = [(self.d_model,), (self.d_model,), (HIDDEN, self.d_model), (HIDDEN,), (HIDDEN,), (HIDDEN,),
                (HIDDEN, HIDDEN), (HIDDEN,), (HIDDEN,), (HIDDEN,), (self.num_classes, HIDDEN), (self.num_classes,)]
        for k, a, w in zip(STATE_KEYS, arrs, want):
            if a.shape != w:
                raise RuntimeError(f"size mismatch for {k}: got {a.shape}, expected {w}")
        import torch
        with torch.cuda.device(self.device):
            ptrs = [a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for a in arrs]
            _lib.check(_lib.lib().prag_prober_load_layer(self._h, int(layer_idx), *ptrs))

    def effective_state_dict(self, layer_idx: int) -> dict:
        """The weights the kernels compute with, as an ImprovedProbe state dict
        whose LayerNorm affines are identity (they are folded into W/b)."""
        d, H, C = self.d_model, HIDDEN, self.num_classes
        W1, b1 = np.empty((H, d), np.float32), np.empty((H,), np.float32)
        W2, b2 = np.empty((H, H), np.float32), np.empty((H,), np.float32)
        W3, b3 = np.empty((C, H), np.float32), np.empty((C,), np.float32)
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        _lib.check(_lib.lib().prag_prober_effective_weights(self._h, int(layer_idx), fp(W1), fp(b1), fp(W2),
                                                            fp(b2), fp(W3), fp(b3)))
        one = lambda n: np.ones((n,), np.float32)
        zero = lambda n: np.zeros((n,), np.float32)
        return {"layer_norm_input.weight": one(d), "layer_norm_input.bias": zero(d),
                "fc1.weight": W1, "fc1.bias": b1,
                "layer_norm1.weight": one(H), "layer_norm1.bias": zero(H),
                "fc2.weight": W2, "fc2.bias": b2,
                "layer_norm2.weight": one(H), "layer_norm2.bias": zero(H),
                "fc3.weight": W3, "fc3.bias": b3}

    def reserve(self, max_batch: int):
        _lib.check(_lib.lib().prag_prober_reserve(self._h, int(max_batch)))

    # -- compute --------------------------------------------------------------
    def _check_x(self, x, ndim):
        import torch
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise RuntimeError("Expected all tensors to be on the same device: the prober lives on "
                               f"{self.device}, got {getattr(x, 'device', type(x))}")
        if x.dim() != ndim or x.shape[-1] != self.d_model:
            raise RuntimeError(f"expected activations of shape [{'L,' if ndim == 3 else ''}B,{self.d_model}], "
                               f"got {tuple(x.shape)}")
        return x if x.is_contiguous() else x.contiguous()

    def forward_layer(self, layer_idx: int, x):
        """``prober(input)`` for one layer: x [B,d] -> logits [B,2] float32."""
        import torch
        x = self._check_x(x, 2)
        B = x.shape[0]
        out = torch.empty((B, self.num_classes), dtype=torch.float32, device=x.device)
        if B == 0:
            return out
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().prag_prober_forward(self._h, ctypes.c_void_p(x.data_ptr()), _x_dtype(x),
                                                      0, int(layer_idx), 1, B,
                                                      ctypes.c_void_p(out.data_ptr()),
                                                      _lib.current_stream_ptr(x.device)))
        return out

    def forward(self, x):
        """All layers in one launch: x [L,B,d] -> logits [L,B,2]."""
        import torch
        x = self._check_x(x, 3)
        L, B = x.shape[0], x.shape[1]
        if L != self.n_layers:
            raise RuntimeError(f"expected {self.n_layers} layers of activations, got {L}")
        out = torch.empty((L, B, self.num_classes), dtype=torch.float32, device=x.device)
        if B == 0:
            return out
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().prag_prober_forward(self._h, ctypes.c_void_p(x.data_ptr()), _x_dtype(x),
                                                      B * self.d_model, 0, L, B,
                                                      ctypes.c_void_p(out.data_ptr()),
                                                      _lib.current_stream_ptr(x.device)))
        return out

    __call__ = forward

    def gate(self, x, ablation: int = 0, threshold: float = 0.0, out=None):
        """exp_rag.py:406-415 for a batch: returns (logits [L,B,2], probsum [B,2],
        decision int32 [B]); decision 1 = retrieve.  ``out`` may carry
        preallocated (logits, probsum, decision) tensors."""
        import torch
        x = self._check_x(x, 3)
        L, B = x.shape[0], x.shape[1]
        if L != self.n_layers:
            raise RuntimeError(f"expected {self.n_layers} layers of activations, got {L}")
        if out is None:
            logits = torch.empty((L, B, 2), dtype=torch.float32, device=x.device)
            probsum = torch.empty((B, 2), dtype=torch.float32, device=x.device)
            decision = torch.empty((B,), dtype=torch.int32, device=x.device)
        else:
            logits, probsum, decision = out
        if B == 0:
            return logits, probsum, decision
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().prag_gate(self._h, ctypes.c_void_p(x.data_ptr()), _x_dtype(x),
                                            B * self.d_model, B, int(ablation), float(threshold),
                                            ctypes.c_void_p(logits.data_ptr()),
                                            ctypes.c_void_p(probsum.data_ptr()),
                                            ctypes.c_void_p(decision.data_ptr()),
                                            _lib.current_stream_ptr(x.device)))
        return logits, probsum, decision

    def decide(self, x, ablation: int = 0, threshold: float = 0.0, with_probsum: bool = False):
        """The gate as the loop consumes it (exp_rag.py:393, 406-415: a host `if` on the two sums): x [L,B,d] on the
        device -> decisions as a host int32 array [B] (1 = retrieve), in ONE C call - gate, copy-out and wait
        (``prag_gate_decide``; nothing is allocated on the device per call).  ``with_probsum`` also returns the sums
        float32 [B,2] the reference prints (exp_rag.py:420)."""
        x = self._check_x(x, 3)
        L, B = x.shape[0], x.shape[1]
        if L != self.n_layers:
            raise RuntimeError(f"expected {self.n_layers} layers of activations, got {L}")
        buf = self._decide_buf
        if buf is None or buf[0].shape[0] < B:
            buf = self._decide_buf = (np.empty((max(B, 8),), np.int32), np.empty((max(B, 8), 2), np.float32))
        dec, ps = buf
        import torch
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().prag_gate_decide(self._h, x.data_ptr(), _x_dtype(x), B * self.d_model, B, int(ablation),
                                                   float(threshold), dec.ctypes.data,
                                                   ps.ctypes.data if with_probsum else None,
                                                   torch.cuda.current_stream(x.device).cuda_stream))
        return (dec[:B].copy(), ps[:B].copy()) if with_probsum else dec[:B].copy()

    _decide_buf = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().prag_prober_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gate_from_logits(logits, ablation: int = 0, threshold: float = 0.0):
    """The gate arithmetic alone on device logits [L,B,2] (or a list of L
    [B,2] tensors, as return_prober_logit_gemma_2b produces)."""
    import torch
    if isinstance(logits, (list, tuple)):
        logits = torch.stack([t.reshape(-1, 2) for t in logits])
    if not logits.is_cuda:
        raise RuntimeError("gate_from_logits needs device logits; there is no CPU path")
    logits = logits.contiguous().float()
    L, B = logits.shape[0], logits.shape[1]
    probsum = torch.empty((B, 2), dtype=torch.float32, device=logits.device)
    decision = torch.empty((B,), dtype=torch.int32, device=logits.device)
    with torch.cuda.device(logits.device):
        _lib.check(_lib.lib().prag_gate_from_logits(ctypes.c_void_p(logits.data_ptr()), L, B, int(ablation),
                                                    float(threshold), ctypes.c_void_p(probsum.data_ptr()),
                                                    ctypes.c_void_p(decision.data_ptr()),
                                                    _lib.current_stream_ptr(logits.device)))
    return probsum, decision


class _LayerView:
    """One layer of an ensemble, callable like the reference's prober."""

    def __init__(self, ens: HipProberEnsemble, layer_idx: int):
        self._ens, self._l = ens, layer_idx

    def __call__(self, x):
        return self._ens.forward_layer(self._l, x)

    def load_state_dict(self, sd, strict: bool = True):
        self._ens.load_layer(self._l, sd)
        return self

    def state_dict(self):
        return self._ens.effective_state_dict(self._l)

    def eval(self):   # inference only: dropout is identity (utils.py:329)
        return self

    def to(self, device=None, *a, **k):
        return self

    def train(self, mode: bool = True):
        """``probe.train()`` (train.py:255-257): this object is the eval-mode forward (dropout =
        identity, utils.py:329).  The training step lives in ``HipProberTrainer`` (train.py:126-135,
        210-220 -> prag_trainer_*), which also hands its parameters back as a state dict."""
        if mode:
            raise RuntimeError("HipProber is the eval-mode prober (utils.py:329); train with "
                               "probing_rag_amd.HipProberTrainer(d_model, ...) / method_2_train and load its "
                               "state_dict() here")
        return self


class HipProber(_LayerView):
    """Drop-in for ``ImprovedProbe(input_size, output_size)`` in eval mode."""

    def __init__(self, input_size: int, output_size: int = 2, hidden_size: int = HIDDEN, weights: str = "f32",
                 device=None):
        if hidden_size != HIDDEN:
            raise NotImplementedError("hidden_size is fixed at 512 (utils.py:30)")
        super().__init__(HipProberEnsemble(1, input_size, output_size, weights=weights, device=device), 0)

    forward = _LayerView.__call__


# ---------------------------------------------------------------------------
# the reference's module-level helpers, same names and argument meaning
# ---------------------------------------------------------------------------
class Config_Maker:
    """utils.py:282-290."""

    def __init__(self, model, method, layer, position, device):
        self.method = method
        self.layer = layer
        self.position = position
        self.device = device
        self.d_model = model.cfg.d_model
        self.model_id = model.cfg.tokenizer_name
        self.num_classes = 2


def load_prober_cfg_gemma_2b(model, config, position, device, start, end, step):
    """utils.py:382-383 (exp_rag.py:311 calls it with 6, 17, 2)."""
    return [config(model, "tokens_mean", j, position, device) for j in range(start, end, step)]


def prober_checkpoint_path(_ds, cfg):
    """The ``--ds`` -> checkpoint path table of utils.py:303-326, as a function: returns the path the
    reference would ``torch.load`` for this cfg, or None where the reference loads nothing (its
    ``else: assert '<string>'`` is a no-op, utils.py:328, so an unknown ``model_id`` silently keeps the
    randomly initialised prober; here such a prober stays unloaded and its first call raises)."""
    m, l, p = cfg.method, cfg.layer, cfg.position
    if cfg.model_id == "mistralai/Mistral-7B-Instruct-v0.1":
        return f"ckpt/probing_ckpt/Mistral-7B-Instruct-v0.1_{m}_probe_2_l{l}_{p}_1.pt"
    if cfg.model_id != "google/gemma-2b":
        return None
    table = {
        25: f"ckpt/_25/0.25_gemma-2b_{m}_2_l{l}_{p}_ep1.pt",
        50: f"ckpt/_5/0.5_gemma-2b_{m}_2_l{l}_{p}_ep1.pt",
        75: f"ckpt/_75/0.75_gemma-2b_{m}_2_l{l}_{p}_ep1.pt",
        777: f"ckpt/_75_full/0.75_gemma-2b_{m}_2_l{l}_{p}_ep.pt",
        3: f"ckpt/_3/in3_1.0_gemma-2b_{m}_2_l{l}_{p}_ep1.pt",
        333: f"ckpt/_3_3/in3_0.33_gemma-2b_{m}_2_l{l}_{p}_ep.pt",
        366: f"ckpt/_3_6/in3_0.66_gemma-2b_{m}_2_l{l}_{p}_ep.pt",
        3000: f"ckpt/_3_1000/in3_1000_gemma-2b_{m}_2_l{l}_{p}_ep11.pt",
        1000: f"ckpt/_1000/1000_gemma-2b_{m}_2_l{l}_{p}_ep11.pt",
    }
    return table.get(_ds, f"ckpt/prob_model_cot_v1/gemma-2b_linear995_{m}_probe_2_l{l}_{p}_1.pt")


def _cfg_device(cfg):
    return cfg.device if str(cfg.device) not in ("cuda", "cpu") else None


def load_prober(_ds, cfg, weights: str = "f32"):
    """utils.py:291-330: ``ImprovedProbe(cfg.d_model, cfg.num_classes).to(cfg.device)``, then
    ``load_state_dict(torch.load(<path chosen by cfg.model_id and --ds>))`` (paths relative to the
    working directory, like the reference's), then ``.eval()``."""
    import torch
    prober = HipProber(cfg.d_model, cfg.num_classes, weights=weights, device=_cfg_device(cfg))
    path = prober_checkpoint_path(_ds, cfg)
    if path is not None:
        prober.load_state_dict(torch.load(path, map_location="cpu"))
    return prober.eval()


def load_prober_models(_ds, cfg_list, weights: str = "f32"):
    """utils.py:385-387, ``[load_prober(_ds, cfg) for cfg in cfg_list]`` - exp_rag.py:312 calls it
    with the ``--ds`` integer.  The probers of all layers live in ONE ensemble
    (``probers[0]._ens``) so the gate can run them in a single launch; each list element still
    behaves like its own ``ImprovedProbe``.  Besides the reference's integer, ``_ds`` may be a list
    of state dicts / checkpoint paths (one per cfg) or a callable ``cfg -> state dict | path``
    (checkpoints are not shipped with the reference)."""
    import torch
    ens = HipProberEnsemble(len(cfg_list), cfg_list[0].d_model, cfg_list[0].num_classes, weights=weights,
                            device=_cfg_device(cfg_list[0]))
    for i, cfg in enumerate(cfg_list):
        if isinstance(_ds, int):
            sd = prober_checkpoint_path(_ds, cfg)
            if sd is None:
                continue                      # the reference loads nothing for this model_id
        else:
            sd = _ds(cfg) if callable(_ds) else _ds[i]
        if isinstance(sd, (str, bytes)):
            sd = torch.load(sd, map_location="cpu")
        ens.load_layer(i, sd)
    return ens.probers


def return_prober_logit_gemma_2b(method_function, cfg_list, model_list):
    """utils.py:389-390, verbatim semantics: one call per layer, logits to CPU."""
    return [method_function(cfg, model).to("cpu") for cfg, model in zip(cfg_list, model_list)]


def _profile_read(fn, handle, cap=4096):
    ms = (ctypes.c_float * cap)()
    n = ctypes.c_int(0)
    _lib.check(fn(handle, ms, cap, ctypes.byref(n)))
    return [float(ms[i]) for i in range(n.value)]


def _ens_profile(self, slots: int):
    """Record HIP events around every fused prober kernel (0 disables)."""
    _lib.check(_lib.lib().prag_prober_profile(self._h, int(slots)))


def _ens_profile_read(self):
    """Durations (ms) of the fused prober kernels recorded since the last read."""
    return _profile_read(_lib.lib().prag_prober_profile_read, self._h)


HipProberEnsemble.profile = _ens_profile
HipProberEnsemble.profile_read = _ens_profile_read
