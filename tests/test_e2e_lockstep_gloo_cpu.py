"""CPU, world 2 over gloo: the lockstep retrieval protocol of `bench.py --e2e` (bench_e2e._Lockstep) - every
rank contributes its pending query or none, the gathered queries are searched as one replicated batch, each
rank keeps its row; ranks that have finished keep serving until all have.  The index is a stand-in that
scores against a small matrix on the CPU (the real one is the HIP ShardedFlatIndex): what is tested is the
protocol - no deadlock with unequal numbers of retrievals per rank, right row to the right rank, termination."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeIndex:
    def __init__(self, X):
        self.X = torch.from_numpy(X)
        self.calls = 0

    def search(self, q, k):
        self.calls += 1
        sc = ((q[:, None, :] - self.X[None, :, :]) ** 2).sum(-1)
        D, I = torch.topk(sc, k, dim=1, largest=False, sorted=True)
        return D, I


def _worker(rank, world, port, q_out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench_e2e
        rng = np.random.default_rng(5)
        X = rng.standard_normal((200, bench_e2e.D_EMB)).astype(np.float32)
        index = _FakeIndex(X)
        lock = bench_e2e._Lockstep(torch, dist, index, world, rank, torch.device("cpu"), 3)
        # rank 0 retrieves 5 times, rank 1 twice (then keeps serving rank 0's retrievals)
        mine = [X[10 * rank + 7 * j] + np.float32(0.001) for j in range(5 if rank == 0 else 2)]
        got = []
        for q in mine:
            ids, done = lock.step(torch.from_numpy(q[None, :]))
            assert not done
            got.append(ids.tolist())
        done, rounds = False, 0
        while not done:
            ids, done = lock.step(None, finished=True)
            assert ids is None
            rounds += 1
        q_out.put((rank, got, rounds, index.calls))
    finally:
        dist.destroy_process_group()


def test_lockstep_retrieval_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q_out.get(timeout=180) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        got, rounds, calls = res[rank]
        want = [10 * rank + 7 * j for j in range(5 if rank == 0 else 2)]
        assert [g[0] for g in got] == want            # each rank got the neighbours of ITS query
    # rank 1 served rank 0's three remaining retrievals, then both saw "everyone finished" in the same step
    assert res[1][1] == 4 and res[0][1] == 1
    assert res[0][2] == res[1][2] == 5                # every collective step with a pending query searched once
