"""Host-side mirror of the reference's prober training objects.

Reference (paths relative to /root/reference):
  * ``probe = ImprovedProbe(d_model, num_classes)``; ``AdamW(probe.parameters(), lr=lr)``;
    ``ExponentialLR(optimizer, gamma=0.995)``                      train.py:126-135
  * ``method_2_train(model, optim, scheduler, activations, labels, pred_lens, args)``
                                                                    utils.py:191-197, train.py:210-220
  * ``method_1_train`` (`each_token`, the script's default) / ``method_3_train`` (`last_token`)
                                                                    utils.py:164-173, 213-220; train.py:182-191, 235-243
  * ``torch.save(probe.state_dict(), path)``                        train.py (checkpoint per epoch)

``HipProberTrainer`` owns the parameters and both Adam moments on the GPU and runs the whole
step (forward in train mode, double-softmax cross entropy, backward, AdamW, LR decay) through
libprag.so.  ``state_dict()`` returns the keys/shapes ``ImprovedProbe.load_state_dict`` and
``HipProber.load_state_dict`` expect.
"""
import ctypes

import numpy as np

from . import _lib
from .prober import STATE_KEYS

_FP = ctypes.POINTER(ctypes.c_float)


class HipProberTrainer:
    def __init__(self, d_model: int, num_classes: int = 2, hidden_size: int = 512, lr: float = 1e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01, gamma: float = 0.995,
                 dropout_p: float = 0.1, seed: int = 0, device=None):
        _lib.require_gpu()
        import torch
        self.d_model, self.num_classes, self.hidden_size = int(d_model), int(num_classes), int(hidden_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_trainer_create(ctypes.byref(h), self.d_model, self.hidden_size,
                                                      self.num_classes, lr, betas[0], betas[1], eps, weight_decay,
                                                      gamma, dropout_p, int(seed) & 0xFFFFFFFF))
        self._h = h
        self.training = True

    # -- nn.Module-like surface ------------------------------------------------
    def _shapes(self):
        d, H, C = self.d_model, self.hidden_size, self.num_classes
        return [(d,), (d,), (H, d), (H,), (H,), (H,), (H, H), (H,), (H,), (H,), (C, H), (C,)]

    def load_state_dict(self, state: dict):
        """Initial parameters (torch tensors or arrays, ImprovedProbe keys); resets the optimiser."""
        import torch
        arrs = []
        for k, shp in zip(STATE_KEYS, self._shapes()):
            v = state[k]
            v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            v = np.ascontiguousarray(v, dtype=np.float32)
            if v.shape != shp:
                raise ValueError(f"{k}: expected {shp}, got {v.shape}")
            arrs.append(v)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_trainer_load(self._h, *[a.ctypes.data_as(_FP) for a in arrs]))
        return self

    def state_dict(self) -> dict:
        import torch
        arrs = [np.empty(shp, np.float32) for shp in self._shapes()]
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_trainer_export(self._h, *[a.ctypes.data_as(_FP) for a in arrs]))
        return {k: torch.from_numpy(a) for k, a in zip(STATE_KEYS, arrs)}

    def train(self, mode: bool = True):
        """``probe.train()`` / ``probe.eval()``: in eval mode the dropout of ``step`` is the identity and
        ``step`` is forward-only."""
        self.training = bool(mode)
        _lib.check(_lib.lib().prag_trainer_set_training(self._h, 1 if mode else 0))
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        return self

    @property
    def lr(self) -> float:
        """``optim.param_groups[0]['lr']``: the rate the next step will use."""
        return float(_lib.lib().prag_trainer_lr(self._h))

    @property
    def steps(self) -> int:
        return int(_lib.lib().prag_trainer_steps(self._h))

    def step(self, x, labels):
        """One optimiser step on pooled states x [B,d_model] (cuda float32) and labels [B].
        Returns (loss 0-d cuda tensor, probs [B,num_classes] cuda) of the forward pass.
        In eval mode (``probe.eval()``, train.py:299) only the forward pass runs: no backward, no
        AdamW / scheduler step - a validation loop cannot train on the dev set."""
        import torch
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        if x.dim() != 2 or x.shape[1] != self.d_model:
            raise ValueError(f"expected [B,{self.d_model}], got {tuple(x.shape)}")
        lab = torch.as_tensor(labels)
        if lab.dim() != 1 or lab.shape[0] != x.shape[0]:
            raise ValueError(f"expected {x.shape[0]} labels, got {tuple(lab.shape)}")
        if lab.numel() and (int(lab.min()) < 0 or int(lab.max()) >= self.num_classes):
            # torch.nn.CrossEntropyLoss raises for a class index outside [0, C) (train.py:149-150)
            raise IndexError(f"Target {int(lab.min()) if int(lab.min()) < 0 else int(lab.max())} is out of bounds.")
        lab = lab.to(device=self.device, dtype=torch.int32).contiguous()
        B = x.shape[0]
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        probs = torch.empty((B, self.num_classes), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_trainer_step(self._h, ctypes.c_void_p(x.data_ptr()),
                                                    ctypes.c_void_p(lab.data_ptr()), B,
                                                    ctypes.c_void_p(loss.data_ptr()), ctypes.c_void_p(probs.data_ptr()),
                                                    _lib.current_stream_ptr(self.device)))
        return loss, probs

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().prag_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def method_2_train(model: HipProberTrainer, optim, scheduler, activations, labels, pred_lens, args=None):
    """utils.py:191-197 / train.py:210-220 with the reference's signature: `optim` and `scheduler`
    are accepted and ignored (the trainer owns AdamW + ExponentialLR).  activations [B,T,d] on the
    GPU, pred_lens [B].  Returns (round(loss, 4), lr after the scheduler step) like the reference."""
    from .loop import pool_ragged
    pooled = pool_ragged(activations, pred_lens, mean=True)
    loss, _ = model.step(pooled, labels)
    return round(loss.item(), 4), model.lr


def method_1_train(model: HipProberTrainer, optim, scheduler, activations, labels, pred_lens, args=None):
    """utils.py:164-173 / train.py:182-191 (`--method each_token`, train.py:354's default): every one of the last
    pred_lens[b] tokens is a training row carrying its sequence's label (`_input_tensor_method1`)."""
    from .loop import pool_each_token
    x, new_labels = pool_each_token(activations, pred_lens, labels, strict=True)
    loss, _ = model.step(x, new_labels)
    return round(loss.item(), 4), model.lr


def method_3_train(model: HipProberTrainer, optim, scheduler, activations, labels, pred_lens=None, args=None):
    """utils.py:213-220 / train.py:235-243 (`--method last_token`): the last position of every sequence."""
    from .loop import pool_last_token
    loss, _ = model.step(pool_last_token(activations), labels)
    return round(loss.item(), 4), model.lr
