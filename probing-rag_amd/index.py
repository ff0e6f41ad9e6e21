"""Host-side mirror of the reference's dense-retriever interface.

Reference call sites (paths relative to /root/reference):
  * ``faiss.IndexFlatL2(768)``, ``index.add(emb)``        make_indexer.py:449-455
  * ``faiss.write_index`` / ``faiss.read_index``          make_indexer.py:457, exp_rag.py:248
  * ``index.search(x, k)`` -> (D, I)                       utils.py:374-380
  * ``encode_query`` / ``find_topk_sim`` / ``batch_topk_sim``  utils.py:365-380

``HipFlatIndex`` keeps the corpus in HBM and answers ``search`` with the fused
scan/top-k kernels of libprag.so.  NumPy in -> NumPy out (like faiss); CUDA
tensors in -> CUDA tensors out (no host round trip).
"""
import ctypes
import struct

import numpy as np

from . import _lib

_STORE = {"f32": _lib.PRAG_F32, "fp32": _lib.PRAG_F32, "f16": _lib.PRAG_F16, "fp16": _lib.PRAG_F16}
METRIC_L2, METRIC_IP, METRIC_COS = _lib.METRIC_L2, _lib.METRIC_IP, _lib.METRIC_COS


class HipFlatIndex:
    """Exact brute-force index: ``IndexFlatL2`` / ``IndexFlatIP`` semantics."""

    def __init__(self, d: int, metric="l2", store: str = "f32", capacity: int = 0, device=None):
        _lib.require_gpu()
        import torch
        self.d = int(d)
        self.metric = _lib.metric_id(metric)
        self.store = store
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.is_trained = True
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_index_create(ctypes.byref(h), self.d, self.metric, _STORE[store],
                                                    int(capacity)))
        self._h = h

    @property
    def ntotal(self) -> int:
        return int(_lib.lib().prag_index_ntotal(self._h))

    def _rows_arg(self, x):
        """-> (pointer, n, is_device, keepalive)"""
        import torch
        if isinstance(x, torch.Tensor):
            if x.dim() != 2 or x.shape[1] != self.d:
                raise ValueError(f"expected [n,{self.d}], got {tuple(x.shape)}")
            x = x.contiguous().float()
            if x.is_cuda:
                return ctypes.c_void_p(x.data_ptr()), x.shape[0], 1, x
            x = x.numpy()
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected [n,{self.d}], got {x.shape}")
        return ctypes.c_void_p(x.ctypes.data), x.shape[0], 0, x

    def _out_arg(self, out, B: int, k: int, device):
        """(D float32 [B,k], I int64 [B,k]) on `device`: fresh tensors, or the caller's after a check - the C entry
        points take bare pointers and write B*k elements of each."""
        import torch
        if out is None:
            return (torch.empty((B, k), dtype=torch.float32, device=device),
                    torch.empty((B, k), dtype=torch.int64, device=device))
        D, I = out
        for t, dt, name in ((D, torch.float32, "D"), (I, torch.int64, "I")):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == torch.device(device)):
                raise ValueError(f"out {name}: expected a CUDA tensor on {device}")
            if t.dtype != dt or tuple(t.shape) != (B, k) or not t.is_contiguous():
                raise ValueError(f"out {name}: expected contiguous {dt} [{B},{k}], got {t.dtype} {tuple(t.shape)}")
        return D, I

    def add(self, x):
        import torch
        if isinstance(x, torch.Tensor) and x.is_cuda and x.device != self.device:
            x = x.to(self.device)          # rows on another GPU: the kernels run on the index's device
        ptr, n, is_dev, keep = self._rows_arg(x)
        # on the caller's current stream of the INDEX's device: device rows (an encoder's output, a
        # .float() copy made in _rows_arg) may still be in flight there
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_index_add(self._h, ptr, n, is_dev, _lib.current_stream_ptr(self.device)))
        del keep

    def add_synthetic(self, seed: int, row0: int, n: int):
        """Append rows [row0,row0+n) of the shared counter-based generator
        (oracle_np.synth_rows) without touching the host."""
        import torch
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_index_add_synthetic(self._h, int(seed) & 0xFFFFFFFF, int(row0), int(n)))

    def search(self, x, k: int, id_offset: int = 0, out=None, tagged: bool = False):
        """-> (D float32 [B,k], I int64 [B,k]); -1 / +-FLT_MAX padded if ntotal < k.
        ``out=(D, I)`` lets device callers supply the result tensors.  ``tagged=True`` (row-sharded search
        only): I carries the float32 residual of every score above the row id, for
        ``merge_topk_packed(..., tagged=True)`` - an exchange format, not ids."""
        import torch
        k = int(k)
        fn = _lib.lib().prag_index_search_tagged if tagged else _lib.lib().prag_index_search
        if isinstance(x, torch.Tensor) and x.is_cuda:
            if x.device != self.device:
                raise ValueError(f"queries on {x.device}, index on {self.device}")
            ptr, B, _, keep = self._rows_arg(x)
            D, I = self._out_arg(out, B, k, x.device)
            with torch.cuda.device(self.device):
                _lib.check(fn(self._h, ptr, B, k, int(id_offset), ctypes.c_void_p(D.data_ptr()),
                              ctypes.c_void_p(I.data_ptr()), 1, _lib.current_stream_ptr(x.device)))
            del keep
            return D, I
        ptr, B, _, keep = self._rows_arg(x)
        D = np.empty((B, k), np.float32)
        I = np.empty((B, k), np.int64)
        with torch.cuda.device(self.device):
            _lib.check(fn(self._h, ptr, B, k, int(id_offset), ctypes.c_void_p(D.ctypes.data),
                          ctypes.c_void_p(I.ctypes.data), 0, _lib.current_stream_ptr(self.device)))
        del keep
        return D, I

    # ---- row-sharded search through the C-level exchange (prag_index_set_comm / prag_index_search_sharded) ----
    def set_comm(self, comm_ptr, rank: int, world: int):
        """Hand the index an RCCL communicator (an ncclComm_t as an integer / c_void_p; None with world 1)."""
        _lib.check(_lib.lib().prag_index_set_comm(self._h, ctypes.c_void_p(comm_ptr) if comm_ptr else None,
                                                  int(rank), int(world)))

    def search_sharded(self, x, k: int, id_offset: int = 0, out=None):
        """Local search + ONE RCCL all-gather + merge in one C call; device tensors in, device tensors out."""
        import torch
        k = int(k)
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(self.device)
        if x.device != self.device:
            raise ValueError(f"queries on {x.device}, index on {self.device}")
        ptr, B, _, keep = self._rows_arg(x)
        D, I = self._out_arg(out, B, k, x.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_index_search_sharded(self._h, ptr, B, k, int(id_offset), ctypes.c_void_p(D.data_ptr()),
                                                            ctypes.c_void_p(I.data_ptr()), _lib.current_stream_ptr(x.device)))
        del keep
        return D, I

    def reconstruct_n(self, row0: int = 0, n: int = None) -> np.ndarray:
        n = self.ntotal - row0 if n is None else n
        out = np.empty((n, self.d), np.float32)
        import torch
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().prag_index_reconstruct(self._h, int(row0), int(n),
                                                         ctypes.c_void_p(out.ctypes.data)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().prag_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def IndexFlatL2(d: int, **kw) -> HipFlatIndex:
    """``faiss.IndexFlatL2(d)`` (make_indexer.py:450)."""
    return HipFlatIndex(d, "l2", kw.pop("store", "f32"), **kw)


def IndexFlatIP(d: int, **kw) -> HipFlatIndex:
    return HipFlatIndex(d, "ip", kw.pop("store", "f32"), **kw)


def packed_result_buffer(B: int, k: int, device, n_parts: int = 1):
    """One byte buffer per shard holding D f32 [B,k] then (8-byte aligned) I i64
    [B,k]; returns (buf uint8 [n_parts, stride], stride, i_offset)."""
    import torch
    i_off = (B * k * 4 + 7) // 8 * 8
    stride = i_off + B * k * 8
    return torch.empty((n_parts, stride), dtype=torch.uint8, device=device), stride, i_off


def packed_views(buf_row, B: int, k: int, i_off: int):
    """(D, I) views into one row of a packed buffer."""
    import torch
    D = buf_row[:B * k * 4].view(torch.float32).view(B, k)
    I = buf_row[i_off:i_off + B * k * 8].view(torch.int64).view(B, k)
    return D, I


def merge_topk_packed(buf, B: int, k: int, metric, tagged: bool = False) -> tuple:
    """Merge the shards of a packed buffer [n_parts, stride] (after ONE all-gather).  ``tagged``: the
    buffer was written by ``search(..., tagged=True)`` - float32 ties across shards are then broken by the
    float64 scores, as in an unsharded search."""
    import torch
    _lib.require_gpu()
    P, stride = buf.shape
    D = torch.empty((B, k), dtype=torch.float32, device=buf.device)
    I = torch.empty((B, k), dtype=torch.int64, device=buf.device)
    with torch.cuda.device(buf.device):
        fn = _lib.lib().prag_merge_topk_packed_tagged if tagged else _lib.lib().prag_merge_topk_packed
        _lib.check(fn(ctypes.c_void_p(buf.data_ptr()), stride, P, B, k,
                                                     _lib.metric_id(metric), ctypes.c_void_p(D.data_ptr()),
                                                     ctypes.c_void_p(I.data_ptr()), _lib.current_stream_ptr(buf.device)))
    return D, I


def merge_topk(D_parts, I_parts, k: int, metric) -> tuple:
    """Exchange step of the row-sharded index on device: D_parts/I_parts are
    CUDA tensors [n_parts,B,k] (e.g. the output of an RCCL all-gather)."""
    import torch
    _lib.require_gpu()
    D_parts = D_parts.contiguous()
    I_parts = I_parts.contiguous()
    P, B, kk = D_parts.shape
    assert kk == k and I_parts.shape == D_parts.shape
    D = torch.empty((B, k), dtype=torch.float32, device=D_parts.device)
    I = torch.empty((B, k), dtype=torch.int64, device=D_parts.device)
    with torch.cuda.device(D_parts.device):
        _lib.check(_lib.lib().prag_merge_topk(ctypes.c_void_p(D_parts.data_ptr()), ctypes.c_void_p(I_parts.data_ptr()),
                                              P, B, k, _lib.metric_id(metric), ctypes.c_void_p(D.data_ptr()),
                                              ctypes.c_void_p(I.data_ptr()), _lib.current_stream_ptr(D_parts.device)))
    return D, I


# ---------------------------------------------------------------------------
# faiss IndexFlat file format (make_indexer.py:457 write_index, exp_rag.py:248
# read_index).  [third-party format, restated from faiss's index_write.cpp as
# published; faiss is not installable here so this is unverified against it]:
#   fourcc "IxF2"(L2) | "IxFI"(IP); int32 d; int64 ntotal; int64 dummy(1<<20) x2;
#   uint8 is_trained; int32 metric_type (0 = IP, 1 = L2); uint64 n_floats; float32[]
# ---------------------------------------------------------------------------
_HDR = "<iqqqBi"          # d, ntotal, dummy, dummy, is_trained, metric_type


def write_index(index: HipFlatIndex, path: str, chunk_rows: int = 1 << 16):
    """The call site of ``faiss.write_index(index, path)`` (make_indexer.py:457): the IndexFlat on-disk layout AS DOCUMENTED
    (fourcc, d, ntotal, two reserved words, is_trained, metric, element count, float32 rows) - faiss is not in this image,
    so the format is unpinned against a file faiss itself wrote (tests use a hand-assembled byte fixture).  The rows are
    streamed out in bounded chunks (device -> host -> file): no second copy of the corpus in HBM or host RAM."""
    l2 = index.metric == METRIC_L2
    n, d = index.ntotal, index.d
    with open(path, "wb") as f:
        f.write(b"IxF2" if l2 else b"IxFI")
        f.write(struct.pack(_HDR, d, n, 1 << 20, 1 << 20, 1, 1 if l2 else 0))
        f.write(struct.pack("<Q", n * d))
        for lo in range(0, n, chunk_rows):
            index.reconstruct_n(lo, min(chunk_rows, n - lo)).tofile(f)


def read_index_header(f):
    """Parse the IndexFlat header at the current position of binary file `f`;
    returns (d, ntotal, metric_name) and leaves `f` at the first row."""
    cc = f.read(4)
    if cc not in (b"IxF2", b"IxFI", b"IxFl"):
        raise ValueError(f"not a flat faiss index (fourcc {cc!r})")
    raw = f.read(struct.calcsize(_HDR))
    if len(raw) != struct.calcsize(_HDR):
        raise ValueError("truncated faiss index header")
    d, ntotal, _, _, _, metric_type = struct.unpack(_HDR, raw)
    if metric_type > 1:
        f.read(4)  # metric_arg
    raw = f.read(8)
    if len(raw) != 8:
        raise ValueError("truncated faiss index header")
    (n_floats,) = struct.unpack("<Q", raw)
    if d <= 0 or ntotal < 0 or n_floats != ntotal * d:
        raise ValueError(f"header says {ntotal}x{d} but holds {n_floats} floats")
    if metric_type not in (0, 1):
        raise ValueError(f"metric_type {metric_type}: only inner product (0) and L2 (1) flat indexes are supported")
    return d, ntotal, "l2" if metric_type == 1 else "ip"


def read_index(path: str, store: str = "f32", chunk_rows: int = 1 << 18) -> HipFlatIndex:
    """The call site of ``faiss.read_index(path)`` (exp_rag.py:248) for files in the documented IndexFlatL2 / IndexFlatIP
    layout (see ``write_index``: unpinned against faiss's own output)."""
    with open(path, "rb") as f:
        try:
            d, ntotal, metric = read_index_header(f)
        except ValueError as e:
            raise ValueError(f"{path}: {e}") from None
        ix = HipFlatIndex(d, metric, store, capacity=ntotal)
        done = 0
        while done < ntotal:
            m = min(chunk_rows, ntotal - done)
            buf = f.read(m * d * 4)
            if len(buf) != m * d * 4:
                raise ValueError(f"{path}: truncated after {done + len(buf) // (d * 4)} of {ntotal} rows")
            ix.add(np.frombuffer(buf, dtype=np.float32).reshape(m, d))
            done += m
    return ix


# ---------------------------------------------------------------------------
# the reference's helpers, same names and argument meaning (utils.py:365-380)
# ---------------------------------------------------------------------------
def encode_query(model_retr, query):
    return model_retr.encode(query)


def find_topk_sim(model_retr, query: str, index, k: int):
    """utils.py:374-376: ONE query string; its [d] embedding is unsqueezed to [1,d].  A host array takes the
    reference's route (torch.tensor -> unsqueeze -> np.array); an embedding that is already a CUDA tensor
    (``MeanPoolEncoder``) stays on the device."""
    import torch
    emb = encode_query(model_retr, query)
    if isinstance(emb, torch.Tensor) and emb.is_cuda:
        return index.search(emb.unsqueeze(0), k=k)
    D, I = index.search(torch.as_tensor(emb).unsqueeze(0).numpy(), k=k)
    return D, I


def batch_topk_sim(model_retr, query, index, k: int):
    D, I = index.search(encode_query(model_retr, query), k=k)
    return D, I


def _ix_set_candidate_depth(self, depth: int):
    """Performance knob (results are exact at every depth): keep `depth` (0=default, 8, 16, 32)
    candidates per query through the scan - deeper lists certify more queries on near-duplicate-dense
    corpora without the exact fallback pass."""
    _lib.check(_lib.lib().prag_index_set_candidate_depth(self._h, int(depth)))


def _ix_last_exact_fallbacks(self) -> int:
    """Queries of the most recent search that the certificate could not clear and the exact float64
    scan recomputed (synchronises the current stream)."""
    n = ctypes.c_int(0)
    _lib.check(_lib.lib().prag_index_last_fallbacks(self._h, _lib.current_stream_ptr(self.device), ctypes.byref(n)))
    return n.value


def _ix_stream_wait_scan(self, stream):
    """Make `stream` (a torch.cuda.Stream) wait for the corpus scan of the most recent search on this index - not for
    the search's tail (bound kernel, exact rerank, fallback probes): work launched on `stream` afterwards, e.g. the
    gate of the next batch, runs beside that tail.  The first call only switches the event recording on."""
    _lib.check(_lib.lib().prag_index_stream_wait_scan(self._h, ctypes.c_void_p(stream.cuda_stream)))


def _ix_last_survivors(self) -> dict:
    """Rows the filter of the most recent two-level search handed to the exact rerank (its last query tile):
    {"queries", "total", "per_query", "max_per_query"}; queries = 0 when that search scanned the rows directly."""
    tot, mx, nq = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.lib().prag_index_last_survivors(self._h, _lib.current_stream_ptr(self.device), ctypes.byref(tot),
                                                    ctypes.byref(mx), ctypes.byref(nq)))
    return {"queries": nq.value, "total": tot.value, "per_query": tot.value / max(1, nq.value), "max_per_query": mx.value}


def parse_plan(line: str) -> dict:
    """`key=value` fields of a plan line -> dict (ints where they parse)."""
    out = {}
    import re
    for m in re.finditer(r"(\w+)=((?:[^= ]| (?!\w+=))*)", line):
        v = m.group(2).strip()
        try:
            out[m.group(1)] = int(v)
        except ValueError:
            out[m.group(1)] = v
    return out


def plan_search(d: int, metric, store: str, ntotal: int, B: int, k: int, shadow: int = 0, n_cu: int = 256) -> dict:
    """The plan `prag_index_search` would execute for this shape (pure host logic: works without a GPU)."""
    buf = ctypes.create_string_buffer(768)
    _lib.check(_lib.lib().prag_plan_search(int(d), _lib.metric_id(metric), _lib.PRAG_F32 if store == "f32" else _lib.PRAG_F16,
                                           int(ntotal), int(B), int(k), int(shadow), int(n_cu), buf, len(buf)))
    return parse_plan(buf.value.decode())


def _ix_last_plan(self) -> dict:
    """The plan the most recent search on this index executed (kernel family, tile, passes, bytes per pass, ...)."""
    buf = ctypes.create_string_buffer(768)
    _lib.check(_lib.lib().prag_index_last_plan(self._h, buf, len(buf)))
    return parse_plan(buf.value.decode())


def _ix_last_tiled8(self) -> int:
    """Batches of > 128 queries on an index that keeps the 8-bit shadow: queries of the most recent search that
    failed the certificate of the int8-tile selection and sent the batch to the fp16 tiles (0: the first tier
    answered; -1: that search did not take the int8 tiles)."""
    n = ctypes.c_int(0)
    _lib.check(_lib.lib().prag_index_last_tiled8(self._h, ctypes.byref(n)))
    return n.value


def _ix_set_shadow(self, mode=1):
    """Two-level exact search through an 8-bit shadow of the rows (prag_index_set_shadow): 0/False off,
    1/True (the default) for shards of >= 2^20 rows when the device has room, 2 at any size.  Same
    results, about half the scan time for batches of <= 64 queries on fp16 rows (a quarter of the bytes
    of float32 rows)."""
    _lib.check(_lib.lib().prag_index_set_shadow(self._h, int(mode)))


def _ix_prepare(self):
    """Build whatever the current mode derives from the rows (the 8-bit shadow) now, on the current
    stream, instead of inside the next search.  ``add`` already does this; needed only after
    ``set_shadow`` switched the shadow on for rows added while it was off."""
    import torch
    with torch.cuda.device(self.device):
        _lib.check(_lib.lib().prag_index_prepare(self._h, _lib.current_stream_ptr(self.device)))


def _ix_reserve(self, B: int, k: int):
    """Allocate the workspaces of a ``B``-query, top-``k`` search now (throw-away searches of a fixed query pattern): later
    searches of that shape allocate nothing and can be captured into a graph from the first call."""
    import torch
    with torch.cuda.device(self.device):
        _lib.check(_lib.lib().prag_index_reserve(self._h, int(B), int(k), _lib.current_stream_ptr(self.device)))


def _ix_set_adaptive(self, on: bool = True):
    """``set_adaptive(False)``: the deterministic plan (prag_index_set_adaptive) - nothing a search launches depends on
    how earlier searches on this handle went or how long they took, so every rank and every run of one input issues
    the same launches; ``last_plan()["adaptive"]`` says which mode ran.  Same results either way."""
    _lib.check(_lib.lib().prag_index_set_adaptive(self._h, 1 if on else 0))


HipFlatIndex.set_adaptive = _ix_set_adaptive
HipFlatIndex.set_shadow = _ix_set_shadow
HipFlatIndex.prepare = _ix_prepare
HipFlatIndex.reserve = _ix_reserve
HipFlatIndex.set_candidate_depth = _ix_set_candidate_depth
HipFlatIndex.last_exact_fallbacks = _ix_last_exact_fallbacks
HipFlatIndex.last_tiled8 = _ix_last_tiled8
HipFlatIndex.last_plan = _ix_last_plan
HipFlatIndex.last_survivors = _ix_last_survivors
HipFlatIndex.stream_wait_scan = _ix_stream_wait_scan


def _ix_set_scan_workgroups(self, n: int):
    """Cap the CUs the (HBM-bound) scan occupies so another stream's kernel can run beside it."""
    _lib.check(_lib.lib().prag_index_set_scan_workgroups(self._h, int(n)))


HipFlatIndex.set_scan_workgroups = _ix_set_scan_workgroups


def _ix_profile(self, slots: int):
    """Record HIP events around every scan_topk launch (0 disables)."""
    _lib.check(_lib.lib().prag_index_profile(self._h, int(slots)))


def _ix_profile_read(self):
    from .prober import _profile_read
    return _profile_read(_lib.lib().prag_index_profile_read, self._h)


def _ix_profile_read_exchange(self):
    """ms per all-gather of the sharded searches since ``profile(slots)`` (C-level exchange only)."""
    from .prober import _profile_read
    return _profile_read(_lib.lib().prag_index_profile_read_exchange, self._h)


HipFlatIndex.profile = _ix_profile
HipFlatIndex.profile_read = _ix_profile_read
HipFlatIndex.profile_read_exchange = _ix_profile_read_exchange
