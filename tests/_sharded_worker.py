"""Worker for tests/test_gpu_sharded_ranks.py: run under torch.distributed.run with 2 ranks on ONE
GPU (gloo for the exchange, since RCCL refuses two ranks per device) - the real HIP local search,
packed result buffers, all-gather and packed merge of ShardedFlatIndex."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    import probing_rag_amd as pra
    from oracle import oracle_np as onp
    N, d, k = 200_001, 768, 10                      # ragged split: 100001 + 100000 rows
    X = onp.synth_rows(42, 0, N, d)
    X[150_000] = X[7]                                # tie across shards -> lowest id
    Q = onp.synth_rows(7, 0, 70, d)                  # > 64 queries: query-stationary kernel
    Q[0] = X[7]
    for metric in ("l2", "cos"):
        ix = pra.ShardedFlatIndex(d, metric, "f16")
        lo, hi = pra.partition_rows(N, world, rank)
        ix.add_local(X[lo:hi])
        ix.sync()
        assert ix.ntotal == N and ix.id_offset == lo
        D, I = ix.search(torch.from_numpy(Q).cuda(), k)
        whole = pra.HipFlatIndex(d, metric, "f16")
        whole.add(X)
        D0, I0 = whole.search(torch.from_numpy(Q).cuda(), k)
        assert torch.equal(I, I0), (rank, metric)
        assert torch.equal(D, D0), (rank, metric)
        if metric == "l2":
            assert I[0, :2].tolist() == [7, 150_000]
    # the pass as one call per rank: the gate over this rank's slice of the next batch rides in the launch of the local
    # search's bound kernel (prag_search_and_gate, tagged ids), then the same exchange
    from tests.golden import cases
    case = cases.PROBER_CASES[1]
    ens = pra.HipProberEnsemble(case["L"], case["d"], 2, weights="f16")
    for l in range(case["L"]):
        ens.load_layer(l, cases.synth_state(case["wseed"] + l, case["d"]))
    xg = torch.from_numpy(cases.synth_x(case["xseed"] + rank, case["L"], 96, case["d"], 1.0)).cuda().half()
    ix = pra.ShardedFlatIndex(d, "cos", "f16")
    lo, hi = pra.partition_rows(N, world, rank)
    ix.engine.index.set_shadow(2)
    ix.add_local(X[lo:hi])
    ix.sync()
    q64 = torch.from_numpy(Q[:64]).cuda()
    D1, I1 = ix.search(q64, k)
    want = [t.clone() for t in ens.gate(xg, 0, 0.0)]
    (D2, I2), got = ix.search_and_gate(q64, k, ens, xg)
    assert ix.engine.index.last_plan()["family"] == "scan8_kernel"
    assert torch.equal(I1, I2) and torch.equal(D1, D2), rank
    assert all(torch.equal(a, b) for a, b in zip(got, want)), rank
    # the deterministic plan (prag_index_set_adaptive(ix, 0)): every rank issues the same launches, run after run -
    # plans on record identical across ranks (the ragged split differs by one row: `rows` / `bytes` aside) and across
    # repeated passes, whatever the handle's history (a detour through the adaptive mode in between)
    local = ix.engine.index
    local.set_adaptive(False)
    plans = []
    for it in range(12):
        if it == 6:
            local.set_adaptive(True)
            for _ in range(3):
                ix.search_and_gate(q64, k, ens, xg)
            local.set_adaptive(False)
        (D3, I3), _ = ix.search_and_gate(q64, k, ens, xg)
        pl = local.last_plan()
        assert pl["adaptive"] == 0
        plans.append({k_: v for k_, v in pl.items() if k_ not in ("rows", "bytes_per_launch", "ws_bytes", "last_seg_rows")})
        assert torch.equal(I3, I1) and torch.equal(D3, D1), rank
    assert all(p_ == plans[0] for p_ in plans), (rank, plans[0], [p_ for p_ in plans if p_ != plans[0]][:1])
    gathered = [None] * world
    dist.all_gather_object(gathered, plans[0])
    assert all(g == gathered[0] for g in gathered), gathered
    dist.barrier()
    if rank == 0:
        print("SHARDED_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
