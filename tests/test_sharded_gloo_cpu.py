"""CPU, world_size 2, gloo: the N>1 path of the row-sharded index — partition,
global id offsets, all-gather exchange, (score,id) merge — driven with the
oracle standing in for the local-shard engine (tests may use the oracle; the
product's only engine is HIP)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_np as onp


class OracleEngine:
    device = torch.device("cpu")

    def __init__(self, metric):
        self.metric, self.rows = metric, np.zeros((0, 0), np.float32)

    @property
    def ntotal(self):
        return len(self.rows)

    def add(self, x):
        x = np.asarray(x, np.float32)
        self.rows = x if self.rows.size == 0 else np.concatenate([self.rows, x])

    def search(self, q, k, id_offset):
        D, I = onp.flat_search(self.rows, np.asarray(q, np.float32), k, self.metric, id_offset=id_offset)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(self, Dp, Ip, k, metric):
        D, I = onp.merge_topk(list(Dp.numpy()), list(Ip.numpy()), k, metric)
        return torch.from_numpy(D), torch.from_numpy(I)


def _worker(rank, world, port, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import probing_rag_amd as pra
        X = onp.synth_rows(42, 0, 301, 64)      # ragged: 301 rows over 2 ranks
        X[200] = X[17]                          # tie across shards -> lowest id wins
        Q = onp.synth_rows(7, 0, 6, 64)
        res = {}
        for metric in (onp.METRIC_L2, onp.METRIC_IP):
            ix = pra.ShardedFlatIndex(64, metric, engine=OracleEngine(metric))
            ix.add_global(X)
            assert ix.ntotal == 301 and ix.id_offset == (0 if rank == 0 else 151)
            D, I = ix.search(Q, 5)
            res[metric] = (D.numpy(), I.numpy())
        # shard smaller than k: padding must sort last in the merge
        ix = pra.ShardedFlatIndex(64, onp.METRIC_L2, engine=OracleEngine(onp.METRIC_L2))
        ix.add_local(X[:2] if rank == 0 else X[2:9])
        with pytest.raises(RuntimeError):
            ix.search(Q, 5)                      # sync() not yet run
        ix.sync()
        D, I = ix.search(Q, 5)
        res["small"] = (D.numpy(), I.numpy())
        q_out.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_sharded_search_equals_unsharded_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q_out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X = onp.synth_rows(42, 0, 301, 64)
    X[200] = X[17]
    Q = onp.synth_rows(7, 0, 6, 64)
    for metric in (onp.METRIC_L2, onp.METRIC_IP):
        D0, I0 = onp.flat_search(X, Q, 5, metric)
        for rank in (0, 1):
            D, I = got[rank][metric]
            assert np.array_equal(I, I0) and np.array_equal(D, D0)
    D0, I0 = onp.flat_search(X[:9], Q, 5, onp.METRIC_L2)
    for rank in (0, 1):
        assert np.array_equal(got[rank]["small"][1], I0)
        assert np.array_equal(got[rank]["small"][0], D0)
