"""Row-sharded flat index: one process per GPU, corpus rows partitioned
contiguously across ranks, local fused top-k, then ONE exchange step — an
all-gather of the per-rank (D, I) [B,k] blocks over RCCL (xGMI) — and a
(score, id) merge on every rank.

The exchange goes through ``torch.distributed.all_gather_into_tensor`` (RCCL under the
"nccl" backend, gloo in the CPU tests).  With ``PRAG_C_EXCHANGE=1`` the whole search
is ONE C call instead, ``prag_index_search_sharded``: local search, ``ncclAllGather``
on the caller's stream, merge (include/prag.h; a communicator of the library's own is
created with ``prag_rccl_*`` because torch.distributed does not expose its
ncclComm_t).  That path is opt-in until a multi-GPU run has validated it; asked for
and unavailable, it raises on every rank - it never falls back silently.
``index.exchange`` / ``index.exchange_note`` say which one runs.

The reference has no distributed code (single process, faiss-cpu on one host:
exp_rag.py:248, 432-436); this is the MI355X-native scaling of that call.
Payload per rank is B*k*12 bytes (120 KB at B=1000, k=10): the exchange is
latency-bound, the scan time falls as 1/world.
"""
import torch

from . import _lib


def ctypes_void(v):
    import ctypes
    return ctypes.c_void_p(v)


def partition_rows(n_total: int, world: int, rank: int):
    """Contiguous ceil partition: rank r holds [r*ceil(N/W), min(N,(r+1)*ceil(N/W)))."""
    per = -(-n_total // world)
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


class HipEngine:
    """Local shard on this rank's GPU (the only engine the product ships)."""

    def __init__(self, d, metric, store, capacity=0, device=None):
        from .index import HipFlatIndex
        self.index = HipFlatIndex(d, metric, store, capacity=capacity, device=device)
        self.device = self.index.device
        self._ws = {}

    @property
    def ntotal(self):
        return self.index.ntotal

    def add(self, x):
        self.index.add(x)

    def add_synthetic(self, seed, row0, n):
        self.index.add_synthetic(seed, row0, n)

    def search(self, q, k, id_offset):
        q = torch.as_tensor(q)
        if not q.is_cuda:
            q = q.to(self.device)
        return self.index.search(q, k, id_offset=id_offset)

    def merge(self, D_parts, I_parts, k, metric):
        from .index import merge_topk
        return merge_topk(D_parts, I_parts, k, metric)

    # single-collective exchange: results are written straight into one packed
    # buffer per shard (D then I), all-gathered once, merged from the packed form
    def search_packed(self, q, k, id_offset, world: int = 1):
        """Local search straight into this rank's slot-sized packed buffer (D then I).  The buffers of a
        (B, k, world) shape are allocated once and reused: a search allocates nothing (the caller owns
        the results until the next search of the same shape)."""
        q = torch.as_tensor(q)
        if not q.is_cuda:
            q = q.to(self.device)
        buf, D, I, gathered = self.packed_buffers(q, k, world)
        # ids tagged with the float32 residual of their float64 score: float32 ties ACROSS shards are then broken
        # by the float64 values in the merge, as an unsharded search breaks them
        self.index.search(q, k, id_offset=id_offset, out=(D, I), tagged=True)
        return buf, D, I, gathered

    def packed_buffers(self, q, k, world: int = 1):
        """(packed send block, its D / I views, the [world, stride] receive block) of a (B, k, world) shape."""
        from .index import packed_result_buffer, packed_views
        B = q.shape[0]
        key = (B, k, world, q.device)
        ws = self._ws.get(key)
        if ws is None:
            buf, stride, i_off = packed_result_buffer(B, k, q.device)
            gathered = torch.empty((world, stride), dtype=torch.uint8, device=q.device) if world > 1 else None
            ws = (buf, packed_views(buf[0], B, k, i_off), gathered)
            if len(self._ws) > 16:
                self._ws.clear()
            self._ws[key] = ws
        buf, (D, I), gathered = ws
        return buf, D, I, gathered

    def merge_packed(self, gathered, B, k, metric):
        from .index import merge_topk_packed
        return merge_topk_packed(gathered, B, k, metric, tagged=True)


class ShardedFlatIndex:
    def __init__(self, d: int, metric="l2", store: str = "f16", capacity: int = 0, group=None, engine=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.d, self.metric = int(d), _lib.metric_id(metric)
        # `engine` exists so the partition / offset / exchange logic can be driven
        # on CPU (gloo) by the test-suite's oracle; the product never passes it.
        self.engine = engine if engine is not None else HipEngine(d, metric, store, capacity)
        self.id_offset = 0
        self.ntotal = 0
        self._synced = True
        self._comm = None           # ncclComm_t of the C-level exchange (enable_c_exchange)
        # Which exchange runs, and why (bench.py prints both): "none" (one rank), "torch.distributed" or "prag_rccl".
        # The C-level exchange (one C call per sharded search, a communicator of the library's own) is OPT-IN until a
        # multi-GPU run has validated it: PRAG_C_EXCHANGE=1 asks for it and RAISES on every rank when any
        # rank cannot have it; unset / "auto" / "0" exchange through torch.distributed (RCCL under the "nccl" backend).
        import os
        want = os.environ.get("PRAG_C_EXCHANGE", "auto")
        self.exchange = "none" if self.world == 1 else "torch.distributed"
        self.exchange_note = "PRAG_C_EXCHANGE=%s: torch.distributed.all_gather_into_tensor (%s)" % (
            want, dist.get_backend(group) if self.distributed else "no process group")
        if engine is None and self.distributed and want == "1":
            ok, why = self.enable_c_exchange()
            if not ok:
                raise RuntimeError(f"PRAG_C_EXCHANGE=1 but the C-level RCCL exchange is unavailable: {why}")

    # ---- exchange in C ---------------------------------------------------------
    def _all_ok(self, ok: int) -> bool:
        """Collective vote: True only if EVERY rank of the group passed `ok` != 0."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.engine.device)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def enable_c_exchange(self):
        """Collective.  Creates an RCCL communicator of the library's own over the ranks of this group (rank 0's
        ncclGetUniqueId travels through torch.distributed) and hands it to the local index; `search` is then one
        C call.  Returns (ok, reason).  The ranks VOTE after every step that can fail on one rank alone - the id,
        ncclCommInitRank, the self-check all-gather - and before the next collective on the new communicator, so no
        rank ever enters a collective that a failed peer skips (a rank whose init failed went to the
        final vote while its peers blocked in the self-check).  torch's own collectives and the library's run on
        different communicators: the device is drained between them (torch.cuda.synchronize) so they never overlap."""
        import ctypes
        lib = _lib.lib()
        why = ""
        ok, comm = 1, ctypes.c_void_p()
        ident = [None]
        try:
            if self.rank == 0:
                buf = ctypes.create_string_buffer(128)
                _lib.check(lib.prag_rccl_unique_id(buf))
                ident = [bytes(buf.raw)]
        except Exception as e:      # noqa: BLE001 - reported, then every rank falls back together
            ok, ident, why = 0, [b""], f"ncclGetUniqueId on rank 0: {e}"
        self.dist.broadcast_object_list(ident, src=self.dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                        group=self.group)
        if len(ident[0]) != 128:
            return False, why or "rank 0 could not create an RCCL unique id (librccl not loadable?)"   # same on every rank
        try:
            with torch.cuda.device(self.engine.device):
                _lib.check(lib.prag_rccl_comm_init_rank(ctypes.byref(comm), self.world, self.rank, ident[0]))
        except Exception as e:  # noqa: BLE001
            ok, why = 0, f"ncclCommInitRank on rank {self.rank}: {e}"
        if not self._all_ok(ok):            # vote BEFORE the first collective on the new communicator
            if comm.value:
                lib.prag_rccl_comm_destroy(comm)
            return False, why or "ncclCommInitRank failed on another rank"
        # the new communicator carries one all-gather before any search relies on it: a wrong answer (or an error
        # from RCCL) is voted on too; it also takes RCCL's one-off channel set-up out of the first search
        torch.cuda.synchronize(self.engine.device)     # torch's all_reduce above is done before ours starts
        try:
            with torch.cuda.device(self.engine.device):
                send = torch.full((4,), 1000 + self.rank, dtype=torch.int32, device=self.engine.device)
                recv = torch.zeros((self.world, 4), dtype=torch.int32, device=self.engine.device)
                _lib.check(lib.prag_rccl_all_gather(comm, send.data_ptr(), recv.data_ptr(), 16,
                                                    torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize(self.engine.device)
                want = (1000 + torch.arange(self.world, dtype=torch.int32, device=self.engine.device))[:, None].expand(-1, 4)
                if not torch.equal(recv, want):
                    raise RuntimeError(f"all-gather self-check returned {recv[:, 0].tolist()}")
        except Exception as e:  # noqa: BLE001
            ok, why = 0, f"RCCL self-check on rank {self.rank}: {e}"
        if not self._all_ok(ok):
            lib.prag_rccl_comm_destroy(comm)
            return False, why or "the RCCL self-check failed on another rank"
        torch.cuda.synchronize(self.engine.device)
        self._comm = comm.value
        self.engine.index.set_comm(self._comm, self.rank, self.world)
        self.exchange = "prag_rccl"
        self.exchange_note = "prag_index_search_sharded: local search + ncclAllGather + merge in one C call (own communicator)"
        return True, ""

    def disable_c_exchange(self):
        """Back to the torch.distributed exchange (collective only in the sense that every rank must do the same)."""
        self.close()
        self.exchange = "none" if self.world == 1 else "torch.distributed"
        self.exchange_note = "torch.distributed.all_gather_into_tensor"

    def close(self):
        if self._comm:
            self.engine.index.set_comm(None, 0, 1)
            _lib.lib().prag_rccl_comm_destroy(ctypes_void(self._comm))
            self._comm = None

    # ---- build -------------------------------------------------------------
    def add_local(self, x):
        """Append rows to THIS rank's shard (call sync() once all ranks are done)."""
        self.engine.add(x)
        self._synced = False

    def add_synthetic_local(self, seed: int, row0: int, n: int):
        self.engine.add_synthetic(seed, row0, n)
        self._synced = False

    def add_global(self, x):
        """Every rank passes the same full [N,d] array; each keeps its partition."""
        lo, hi = partition_rows(len(x), self.world, self.rank)
        self.engine.add(x[lo:hi])
        self._synced = False
        self.sync()

    def sync(self):
        """Collective: global row id of a shard's first row = rows on lower ranks."""
        n_local = int(self.engine.ntotal)
        if self.distributed and self.world > 1:
            dev = getattr(self.engine, "device", torch.device("cpu"))
            counts = torch.zeros(self.world, dtype=torch.int64, device=dev)
            mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
            self.dist.all_gather_into_tensor(counts, mine, group=self.group)
            counts = counts.cpu().tolist()
        else:
            counts = [n_local]
        self.id_offset = int(sum(counts[:self.rank]))
        self.ntotal = int(sum(counts))
        self._synced = True

    # ---- search ------------------------------------------------------------
    def search(self, q, k: int):
        """q replicated on every rank -> identical (D [B,k], I [B,k]) on every rank."""
        if not self._synced:
            raise RuntimeError("ShardedFlatIndex.sync() must run (on every rank) after adding rows")
        if self._comm is not None:      # local search + ncclAllGather + merge: one C call
            return self.engine.index.search_sharded(q, k, self.id_offset)
        if hasattr(self.engine, "search_packed"):
            if self.world == 1:     # nothing to exchange: plain local search, results owned by the caller
                return self.engine.search(q, k, self.id_offset)
            buf, D_loc, I_loc, gathered = self.engine.search_packed(q, k, self.id_offset, self.world)
            self.dist.all_gather_into_tensor(gathered, buf, group=self.group)
            return self.engine.merge_packed(gathered, D_loc.shape[0], k, self.metric)
        D_loc, I_loc = self.engine.search(q, k, self.id_offset)
        if self.world == 1:
            return D_loc, I_loc
        B = D_loc.shape[0]
        # concatenated-along-dim-0 form (valid on both RCCL and gloo); rank r's block is rows [r*B,(r+1)*B)
        D_all = torch.empty((self.world * B, k), dtype=D_loc.dtype, device=D_loc.device)
        I_all = torch.empty((self.world * B, k), dtype=I_loc.dtype, device=I_loc.device)
        self.dist.all_gather_into_tensor(D_all, D_loc.contiguous(), group=self.group)
        self.dist.all_gather_into_tensor(I_all, I_loc.contiguous(), group=self.group)
        return self.engine.merge(D_all.view(self.world, B, k), I_all.view(self.world, B, k), k, self.metric)


def _sharded_search_and_gate(self, q, k: int, ens, x, ablation: int = 0, threshold: float = 0.0, gate_out=None):
    """``search(q, k)`` AND ``ens.gate(x, ablation, threshold)`` over this rank's slice of the NEXT batch of pooled states
    (gate rows are split across ranks, replicas only): the gate's prober workgroups ride in the launch of the LOCAL
    search's bound kernel (``prag_search_and_gate``), then the usual exchange - one all-gather of the packed local
    top-k, the (score, residual, id) merge.  Returns ((D, I), (logits, probsum, decision))."""
    from .loop import search_and_gate
    if not self._synced:
        raise RuntimeError("ShardedFlatIndex.sync() must run (on every rank) after adding rows")
    q = torch.as_tensor(q)
    if not q.is_cuda:
        q = q.to(self.engine.device)
    if self.world == 1:
        return search_and_gate(self.engine.index, q, k, ens, x, ablation, threshold, gate_out=gate_out, id_offset=self.id_offset)
    if self._comm is not None:          # the exchange in C has no fused form: the search, then the gate
        out = self.engine.index.search_sharded(q, k, self.id_offset)
        return out, ens.gate(x, ablation, threshold, out=gate_out)
    buf, D_loc, I_loc, gathered = self.engine.packed_buffers(q, k, self.world)
    _, gate_out = search_and_gate(self.engine.index, q, k, ens, x, ablation, threshold, out=(D_loc, I_loc), gate_out=gate_out,
                                  id_offset=self.id_offset, tagged=True)
    self.dist.all_gather_into_tensor(gathered, buf, group=self.group)
    return self.engine.merge_packed(gathered, D_loc.shape[0], k, self.metric), gate_out


ShardedFlatIndex.search_and_gate = _sharded_search_and_gate


def search_shards_on_one_gpu(shards, q, k: int, metric, packed: bool = True):
    """1-GPU shard simulation (gpurun gives one GPU): `shards` is a list of
    HipFlatIndex holding consecutive row ranges; runs the same local-search +
    merge code the multi-GPU path runs (packed single-buffer exchange format by
    default), without the collective."""
    from .index import merge_topk, merge_topk_packed, packed_result_buffer, packed_views
    B, off = q.shape[0], 0
    if packed:
        buf, stride, i_off = packed_result_buffer(B, k, q.device, n_parts=len(shards))
        for p, ix in enumerate(shards):
            ix.search(q, k, id_offset=off, out=packed_views(buf[p], B, k, i_off), tagged=True)
            off += ix.ntotal
        return merge_topk_packed(buf, B, k, metric, tagged=True)
    Ds, Is = [], []
    for ix in shards:
        D, I = ix.search(q, k, id_offset=off)
        Ds.append(D)
        Is.append(I)
        off += ix.ntotal
    return merge_topk(torch.stack(Ds), torch.stack(Is), k, metric)
