"""The retrieve-decide loop and its input producer, restated for the HIP path.

Reference (paths relative to /root/reference):
  * hook_fn / add_layer_hook            exp_rag.py:315-329  (activations.detach().cpu() per pass)
  * return_mean_output                  exp_rag.py:381-389  (cat(cache[1:],1).sum(1) -> prober)
  * the per-query loop                  exp_rag.py:396-474  (<= 4 retrieval rounds)
  * return_evidences                    exp_rag.py:369-379

``HiddenStatePool`` replaces the hook cache with a running on-device
accumulator: no D2H copy per token, no H2D before the
gate.  ``retrieve_decide`` is the loop with the model / prompt / corpus pieces
injected, so exp_rag.py's control flow can be driven unchanged.
"""
import ctypes

from . import _lib


_DT = None


def _DTYPES():
    global _DT
    if _DT is None:
        import torch
        _DT = {torch.float32: _lib.PRAG_F32, torch.float16: _lib.PRAG_F16, torch.bfloat16: _lib.PRAG_BF16}
    return _DT


class HiddenStatePool:
    """acc[l] = sum over all positions of every forward pass after the first.

    Every hook call adds its activations at once, in its own launch (``defer=False``, the default: the tensor is read
    while the model still holds it unchanged, whatever the model does with that buffer afterwards).
    ``defer=True`` adds a decode step (one position per pass, KV cache on) of ALL hooked layers in ONE launch
    (``prag_pool_accumulate_layers``): each layer's hook only notes where the model left its activations, the last
    layer's hook - or ``pooled()`` - flushes the set.  The noted tensors are kept alive until then and must not be
    overwritten in place meanwhile: TransformerLens hook points and HF decoder-layer outputs are fresh tensors (the
    call sites that opt in: bench_e2e.py, tests), static / compiled output buffers or an in-place residual add are
    not - a tensor whose version counter moved between the note and the flush raises instead of being summed wrong.

    ``attach_gate(ens, ablation, threshold)`` (round 6, with ``defer=True``): the launch that adds a decode step also runs
    the gate on the sums it has just formed (``prag_pool_step_gate``) and leaves the decision in pinned host memory, so
    ``pool.decide()`` - exp_rag.py:393, 406-415's host branch - finds the decision of the last step already there when
    `generate` returns instead of starting three launches on a host the LM has left cold (5 us against 90-120 in
    the loop).  Same arithmetic as ``ens.decide(pool.pooled())``: bit-identical sums and decisions; shapes outside the
    small-batch gate (batch > 4) fall back to exactly that call."""

    def __init__(self, n_layers: int, d_model: int, batch: int = 1, device=None, defer: bool = False):
        _lib.require_gpu()
        import torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # two buffers: the step launch reads the sums of one while it writes the other (a layer's workgroups all read
        # the old sums); every other path works in place on the current one
        self._acc2 = torch.zeros((2, n_layers, batch, d_model), dtype=torch.float32, device=self.device)
        self._cur = 0
        self.passes = [0] * n_layers
        self.defer = bool(defer) and n_layers <= 64
        self._pending = {}      # slot -> (tensor [B,1,d], assign, tensor._version when noted)
        self._gate = None       # (ens, ablation, threshold) of attach_gate
        self._step_tag = None   # tag of the step launch whose decision describes the CURRENT sums, else None
        self._step_ok = True    # False once the library has said the shape is outside the step launch

    @property
    def acc(self):
        """float32 [L, B, d_model]: the running sums (a view of the current one of the two buffers)."""
        return self._acc2[self._cur]

    def attach_gate(self, ens, ablation: int = 0, threshold: float = 0.0):
        """From now on a decode step's launch also decides (see the class docstring); ``ens`` is a
        ``HipProberEnsemble`` with every layer loaded, on this pool's device.  ``attach_gate(None)`` detaches."""
        if ens is not None and (ens.n_layers != len(self.passes) or ens.d_model != self._acc2.shape[-1]):
            raise ValueError(f"the pool holds {len(self.passes)} layers x {self._acc2.shape[-1]}, the ensemble "
                             f"{ens.n_layers} x {ens.d_model}")
        self._gate = None if ens is None else (ens, int(ablation), float(threshold))
        self._step_tag, self._step_ok = None, True
        # decide() runs on a host the LM has left cold: everything it needs is made here, once
        import numpy as np
        self._dec_buf = np.empty((8,), np.int32)
        self._ps_buf = np.empty((8, 2), np.float32)
        self._dec_ptr = ctypes.c_void_p(self._dec_buf.ctypes.data)
        self._ps_ptr = ctypes.c_void_p(self._ps_buf.ctypes.data)
        self._step_result_fn = _lib.lib().prag_gate_step_result
        self._step_stream = None
        return self

    def reset(self):            # `cache = {}` exp_rag.py:397, 423: noted-but-unflushed activations are DROPPED with it
        self._pending = {}
        self.passes = [0] * len(self.passes)
        self._step_tag = None

    def _flush(self):
        """Add the noted decode-step activations: one launch when every layer noted one of the same kind."""
        import torch
        if not self._pending:
            return
        pend, self._pending = self._pending, {}
        L = len(self.passes)
        dts = _DTYPES()
        for s_, (a_, _, ver) in pend.items():
            if a_._version != ver:
                raise RuntimeError(f"HiddenStatePool(defer=True): the activations noted for layer slot {s_} were modified "
                                   "in place before they were pooled; use defer=False with this model")
        items = [(pend[s][0], pend[s][1]) if s in pend else None for s in range(L)]
        full = all(it is not None for it in items)
        if full:
            a0, as0 = items[0]
            full = all(a.dtype == a0.dtype and a.shape == a0.shape and a.device == a0.device and asg == as0
                       for a, asg in items) and tuple(a0.shape[:1]) == tuple(self.acc.shape[1:2])
        self._step_tag = None                   # the sums are about to change
        if full:
            a0, as0 = items[0]
            n = a0.numel()
            ptrs = (ctypes.c_void_p * L)(*[a.data_ptr() for a, _ in items])
            with _lib.on_device(a0.device):
                if self._gate is not None and self._step_ok and a0.device == self.device:
                    ens, abl, thr = self._gate
                    tag = ctypes.c_uint64(0)
                    nxt = self._cur ^ 1
                    rc = _lib.lib().prag_pool_step_gate(ens._h, ctypes.c_void_p(self._acc2[self._cur].data_ptr()),
                                                        ctypes.c_void_p(self._acc2[nxt].data_ptr()), ptrs, dts[a0.dtype],
                                                        a0.shape[0], as0, abl, thr, ctypes.byref(tag),
                                                        _lib.current_stream_ptr(a0.device))
                    if rc == _lib.PRAG_OK:
                        self._cur, self._step_tag = nxt, ctypes.c_uint64(tag.value)
                        self._step_stream = _lib.current_stream_ptr(a0.device)     # (what decide() waits on if it has to)
                        return
                    if rc != -4:                 # PRAG_EUNSUPPORTED: this shape never takes the step launch
                        _lib.check(rc)
                    self._step_ok = False
                _lib.check(_lib.lib().prag_pool_accumulate_layers(ctypes.c_void_p(self.acc.data_ptr()), ptrs, L, dts[a0.dtype],
                                                                  n, as0, _lib.current_stream_ptr(a0.device)))
            return
        for slot, it in enumerate(items):
            if it is None:
                continue
            a, asg = it
            with torch.cuda.device(a.device):
                _lib.check(_lib.lib().prag_pool_accumulate(ctypes.c_void_p(self.acc[slot].data_ptr()), ctypes.c_void_p(a.data_ptr()),
                                                           dts[a.dtype], a.numel(), asg, _lib.current_stream_ptr(a.device)))

    def hook(self, slot: int):
        """Returns a forward hook for layer slot `slot` (``model.add_hook(name, fn)``)."""
        def fn(activations, hook=None):
            self.observe(slot, activations)
            return activations
        return fn

    def observe(self, slot: int, activations):
        import torch
        n = self.passes[slot]
        self.passes[slot] = n + 1
        if n == 0:
            return              # element 0 = the prompt pass, skipped by cache[name][1:]
        a = activations.detach()
        if a.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            a = a.float()
        a = a.contiguous()
        Bt, T, d = a.shape
        dst = self.acc[slot]
        dt = {torch.float32: _lib.PRAG_F32, torch.float16: _lib.PRAG_F16, torch.bfloat16: _lib.PRAG_BF16}[a.dtype]
        st = _lib.current_stream_ptr(a.device)
        if T == 1 and self.defer and (Bt * d) % 4 == 0:
            if slot in self._pending:        # a second pass of this layer before the others caught up
                self._flush()
            self._pending[slot] = (a, 1 if n == 1 else 0, a._version)
            if len(self._pending) == len(self.passes):
                self._flush()
            return
        self._flush()                        # keep the order of additions per layer
        self._step_tag = None
        with torch.cuda.device(a.device):
            if T == 1:
                _lib.check(_lib.lib().prag_pool_accumulate(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(a.data_ptr()),
                                                           dt, Bt * d, 1 if n == 1 else 0, st))
            else:   # a pass without KV cache: sum all of its positions first
                lens = torch.full((Bt,), T, dtype=torch.int64, device=a.device)
                tmp = torch.empty((Bt, d), dtype=torch.float32, device=a.device)
                _lib.check(_lib.lib().prag_pool_ragged(ctypes.c_void_p(a.data_ptr()), dt, Bt, T, d,
                                                       ctypes.c_void_p(lens.data_ptr()), 0,
                                                       ctypes.c_void_p(tmp.data_ptr()), st))
                _lib.check(_lib.lib().prag_pool_accumulate(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(tmp.data_ptr()),
                                                           _lib.PRAG_F32, Bt * d, 1 if n == 1 else 0, st))

    def pooled(self):
        self._flush()
        if min(self.passes) < 2:   # torch.concat([]) raises in the reference
            raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")
        return self.acc

    def decide(self, with_probsum: bool = False):
        """exp_rag.py:393, 406-415 on the pooled sums, as a host int32 array [B] (1 = retrieve) - what
        ``ens.decide(pool.pooled(), ablation, threshold)`` returns, for the ensemble given to ``attach_gate``.  When the
        last decode step's launch already decided (nothing was added since) this only reads that result from host
        memory; otherwise it is that call."""
        gate = self._gate
        if gate is None:
            raise RuntimeError("HiddenStatePool.decide() needs attach_gate(ens, ablation, threshold) first")
        ens, abl, thr = gate
        if self._pending:
            self._flush()
        tag = self._step_tag
        if tag is not None:          # (a tag implies every layer has seen a decode step: pooled()'s check is not needed)
            B = self._acc2.shape[2]
            rc = self._step_result_fn(ens._h, tag, B, self._dec_ptr, self._ps_ptr if with_probsum else None, self._step_stream)
            if rc == _lib.PRAG_OK:
                dec = self._dec_buf[:B].copy()
                return (dec, self._ps_buf[:B].copy()) if with_probsum else dec
            if rc != -5:                         # PRAG_ESTATE: overwritten / voided - the sums are intact, decide on them
                _lib.check(rc)
        x = self.pooled()
        return ens.decide(x, abl, thr, with_probsum=with_probsum)


def pool_ragged(acts, pred_lens, mean: bool = True):
    """train.py:153-162 + 202-205 on device: [B,T,d] -> [B,d] over the last
    pred_lens[b] positions (mean) or their sum (inference-time pooling)."""
    import torch
    _lib.require_gpu()
    acts = acts.contiguous()
    if acts.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        acts = acts.float()
    B, T, d = acts.shape
    lens = torch.as_tensor(pred_lens, dtype=torch.int64, device=acts.device).contiguous()
    out = torch.empty((B, d), dtype=torch.float32, device=acts.device)
    with torch.cuda.device(acts.device):
        _lib.check(_lib.lib().prag_pool_ragged(ctypes.c_void_p(acts.data_ptr()),
                                               {torch.float32: _lib.PRAG_F32, torch.float16: _lib.PRAG_F16,
                                                torch.bfloat16: _lib.PRAG_BF16}[acts.dtype],
                                               B, T, d, ctypes.c_void_p(lens.data_ptr()), 1 if mean else 0,
                                               ctypes.c_void_p(out.data_ptr()), _lib.current_stream_ptr(acts.device)))
    return out


def pool_each_token(acts, pred_lens, labels=None, strict: bool = False):
    """`_input_tensor_method1` (train.py:153-162, utils.py:134-143) on device: the last pred_lens[b] positions of every
    sample of acts [B,T,d], concatenated -> float32 [sum(pred_lens), d], and (if `labels` is given)
    ``torch.repeat_interleave(labels, pred_lens)`` as an int64 tensor.  `pred_lens` comes from the data loader (host):
    the row offsets are a host prefix sum, like the reference's Python loop over samples.
    ``strict`` (what method_1_train / method_1_eval pass): a pred_len of 0 or above T raises - in the reference `-0:` takes
    ALL T positions and an over-long pred_len takes T, rows and repeated labels then differ in count and the loss
    raises; without it such lengths are clipped to [0, T] (a ragged gather in its own right)."""
    import numpy as np
    import torch
    _lib.require_gpu()
    acts = acts.contiguous()
    if acts.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        acts = acts.float()
    B, T, d = acts.shape
    lens = np.asarray(torch.as_tensor(pred_lens).cpu(), dtype=np.int64).reshape(-1)
    if lens.shape[0] != B:
        raise ValueError(f"expected {B} pred_lens, got {lens.shape[0]}")
    if strict and ((lens <= 0) | (lens > T)).any():
        bad = int(np.flatnonzero((lens <= 0) | (lens > T))[0])
        raise ValueError(f"pred_lens[{bad}] = {int(lens[bad])} outside [1, {T}]: rows and repeated labels would differ in "
                         "count (utils.py:134-143)")
    lens = np.clip(lens, 0, T)
    off = np.zeros(B + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    n_rows = int(off[-1])
    off_dev = torch.from_numpy(off).to(acts.device)
    out = torch.empty((n_rows, d), dtype=torch.float32, device=acts.device)
    lab_in = lab_out = None
    if labels is not None:
        lab_in = torch.as_tensor(labels).to(device=acts.device, dtype=torch.int32).contiguous()
        lab_out = torch.empty((n_rows,), dtype=torch.int32, device=acts.device)
    with torch.cuda.device(acts.device):
        _lib.check(_lib.lib().prag_pool_each_token(
            ctypes.c_void_p(acts.data_ptr()),
            {torch.float32: _lib.PRAG_F32, torch.float16: _lib.PRAG_F16, torch.bfloat16: _lib.PRAG_BF16}[acts.dtype],
            B, T, d, ctypes.c_void_p(off_dev.data_ptr()), n_rows,
            ctypes.c_void_p(lab_in.data_ptr()) if lab_in is not None else None, ctypes.c_void_p(out.data_ptr()),
            ctypes.c_void_p(lab_out.data_ptr()) if lab_out is not None else None, _lib.current_stream_ptr(acts.device)))
    return out, (lab_out.long() if lab_out is not None else None)


def pool_last_token(acts):
    """`activations[:, -1, :]` (train.py:228, utils.py:206; the `last_token` method) as float32 [B,d] on device:
    the ragged pool over ONE trailing position."""
    import torch
    return pool_ragged(acts, torch.ones((acts.shape[0],), dtype=torch.int64), mean=False)


def _eval_forward(prober, x, labels):
    """make_loss + return_acc (train.py:141-151, 170-181) on pooled rows: softmax(-1), CrossEntropyLoss applied to the
    probabilities (the reference's double softmax), argmax accuracy."""
    import torch
    probs = torch.softmax(prober(x), dim=-1)
    labels = torch.as_tensor(labels, device=probs.device).long()
    loss = torch.nn.functional.cross_entropy(probs, labels)
    correct = (torch.argmax(probs, dim=-1) == labels).sum().item()
    return round(correct / labels.size(0), 4), loss, probs


def method_1_eval(prober, activations, labels, pred_lens, args=None, return_probs: bool = False):
    """train.py:193-197 / utils.py:175-179 (`each_token`): every one of the last pred_len tokens is a sample with its
    sequence's label.  Returns the reference's 3-tuple (accuracy over the TOKEN rows rounded to 4 places, len(labels) -
    the number of sequences, as the reference returns it -, loss); ``return_probs=True`` appends the probabilities.
    `args` (the reference's fifth positional: only `.device` is read there) is accepted and unused."""
    x, new_labels = pool_each_token(activations, pred_lens, labels, strict=True)
    acc, loss, probs = _eval_forward(prober, x, new_labels)
    return (acc, len(labels), loss, probs) if return_probs else (acc, len(labels), loss)


def method_3_eval(prober, activations, labels, pred_lens=None, args=None, return_probs: bool = False):
    """train.py:245-249 / utils.py:222-226 (`last_token`): the prober on the last position only.  Same returns."""
    acc, loss, probs = _eval_forward(prober, pool_last_token(activations), labels)
    return (acc, len(labels), loss, probs) if return_probs else (acc, len(labels), loss)


def method_2_eval(prober, activations, labels, pred_lens, args=None, return_probs: bool = False):
    """train.py:199-208, 222-225 / utils.py:181-203 (`_method_2_util` + `return_acc`), forward
    only: ragged last-`pred_len` MEAN pool (HIP) -> prober (HIP) -> softmax(-1) ->
    CrossEntropyLoss applied to the probabilities (the reference's double softmax,
    train.py:149-150) -> argmax accuracy.  activations [B,T,d] on the GPU.
    Returns (accuracy rounded to 4 places, n, loss) as the reference does; ``return_probs=True`` appends probs."""
    pooled = pool_ragged(activations, pred_lens, mean=True)
    acc, loss, probs = _eval_forward(prober, pooled, labels)
    return (acc, len(labels), loss, probs) if return_probs else (acc, len(labels), loss)


def masked_mean_pool(hidden, attention_mask):
    """Contriever / sentence-transformers mean pooling (utils.py:365-366 ->
    SentenceTransformer.encode, third-party) on device: hidden [B,T,d] (f32/f16/bf16),
    attention_mask [B,T] -> float32 [B,d] embeddings, ready for HipFlatIndex.search."""
    import torch
    _lib.require_gpu()
    hidden = hidden.contiguous()
    dt = {torch.float32: _lib.PRAG_F32, torch.float16: _lib.PRAG_F16, torch.bfloat16: _lib.PRAG_BF16}.get(hidden.dtype)
    if dt is None:
        hidden, dt = hidden.float(), _lib.PRAG_F32
    B, T, d = hidden.shape
    mask = attention_mask.to(device=hidden.device, dtype=torch.int64).contiguous()
    out = torch.empty((B, d), dtype=torch.float32, device=hidden.device)
    with torch.cuda.device(hidden.device):
        _lib.check(_lib.lib().prag_pool_masked_mean(ctypes.c_void_p(hidden.data_ptr()), dt, ctypes.c_void_p(mask.data_ptr()),
                                                    B, T, d, ctypes.c_void_p(out.data_ptr()),
                                                    _lib.current_stream_ptr(hidden.device)))
    return out


def search_and_gate(index, q, k: int, ens, x, ablation: int = 0, threshold: float = 0.0, out=None, gate_out=None,
                    id_offset: int = 0, tagged: bool = False):
    """One pass of the hot path as one C call (``prag_search_and_gate``): ``index.search(q, k)`` (utils.py:379) AND
    ``ens.gate(x, ablation, threshold)`` over the NEXT batch of pooled states (exp_rag.py:406-415) - independent work
    of a loop that keeps batches in flight.  On a two-level search the gate's prober workgroups ride in the launch of the
    search's bound kernel (the search's tail leaves 3/4 of the chip idle); otherwise it is the two calls in a row.
    Same results either way.  q [B,d] and x [L,Bg,d_model] are CUDA tensors; returns ((D, I), (logits, probsum,
    decision)).  ``tagged``: ids in the exchange format of a row-sharded search (``ShardedFlatIndex.search_and_gate``)."""
    import torch
    _lib.require_gpu()
    if not (isinstance(q, torch.Tensor) and q.is_cuda):
        raise RuntimeError("search_and_gate needs device queries (CUDA tensor [B,d])")
    if q.device != index.device:
        raise ValueError(f"queries on {q.device}, index on {index.device}")
    q_ptr, B, _, keep = index._rows_arg(q)        # [B, index.d] float32 contiguous, or ValueError (the C entry has no d)
    x = ens._check_x(x, 3)
    if x.device != index.device:
        raise ValueError(f"pooled states on {x.device}, index on {index.device}")
    L, Bg = x.shape[0], x.shape[1]
    if L != ens.n_layers:
        raise RuntimeError(f"expected {ens.n_layers} layers of activations, got {L}")
    k = int(k)
    out = index._out_arg(out, B, k, index.device)
    if gate_out is None:
        gate_out = (torch.empty((L, Bg, 2), dtype=torch.float32, device=x.device),
                    torch.empty((Bg, 2), dtype=torch.float32, device=x.device),
                    torch.empty((Bg,), dtype=torch.int32, device=x.device))
    else:
        for t, shape, dt, name in ((gate_out[0], (L, Bg, 2), torch.float32, "logits"),
                                   (gate_out[1], (Bg, 2), torch.float32, "probsum"),
                                   (gate_out[2], (Bg,), torch.int32, "decision")):
            if not (isinstance(t, torch.Tensor) and t.device == x.device and t.dtype == dt and tuple(t.shape) == shape
                    and t.is_contiguous()):
                raise ValueError(f"gate_out {name}: expected contiguous {dt} {list(shape)} on {x.device}")
    from .prober import _x_dtype
    with torch.cuda.device(index.device):
        _lib.check(_lib.lib().prag_search_and_gate(
            index._h, q_ptr, B, k, int(id_offset), ctypes.c_void_p(out[0].data_ptr()),
            ctypes.c_void_p(out[1].data_ptr()), ens._h, ctypes.c_void_p(x.data_ptr()), _x_dtype(x), Bg * ens.d_model, Bg,
            int(ablation), float(threshold), ctypes.c_void_p(gate_out[0].data_ptr()), ctypes.c_void_p(gate_out[1].data_ptr()),
            ctypes.c_void_p(gate_out[2].data_ptr()), 1 if tagged else 0, _lib.current_stream_ptr(index.device)))
    del keep
    return out, gate_out


def return_evidences(retrieved_passages) -> str:
    """exp_rag.py:369-379 (dense branch: passages are plain strings)."""
    return "\n".join(f"passage {n + 1}: {p}" for n, p in enumerate(retrieved_passages))


def retrieve_decide(question: str, first_input, *, generate, gate, retrieve, lookup, make_prompt, tokenize,
                    to_string, reset=lambda: None, k: int = 5):
    """One query of exp_rag.py:396-474.

    generate(inputs) -> output ids ; gate() -> 0 (stop) | 1 (retrieve) ;
    retrieve(text, k) -> (D, I) ; lookup(ids) -> list[str] ;
    make_prompt(question, evidences) -> str ; tokenize(str) -> inputs ;
    to_string(output) -> list[str] ; reset() clears the hidden-state pool.
    Returns (prediction text, retr_count) with the reference's cap semantics:
    at most 4 rounds run, retr_count saturates at 3 (exp_rag.py:462-465).
    """
    reset()
    output = generate(first_input)
    retr_count = 0
    decision = gate()
    if decision == 0:
        return to_string(output)[0], retr_count
    search_input_new = None
    while decision == 1:
        reset()
        query = question if retr_count == 0 else search_input_new[0]
        _, I = retrieve(query, k)
        ids = I[0].tolist()
        evidences = return_evidences(lookup(ids))
        output = generate(tokenize(make_prompt(question, evidences)))
        decision = gate()
        search_input_new = to_string(output)
        if retr_count > 2:
            break
        retr_count += 1
    return search_input_new[0], retr_count
