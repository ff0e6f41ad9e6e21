"""Worker for tests/test_gpu_rccl_one_rank.py: ONE rank under the "nccl" backend (= RCCL on ROCm) on the one GPU a
gpurun box has.  No xGMI link is crossed, but RCCL is initialised the way bench.py initialises it (device_id,
HSA_ENABLE_IPC_MODE_LEGACY=0) and runs every collective the row-sharded path issues, with the dtypes and shapes it
issues them in: the packed uint8 all-gather of local top-k lists, the int64 all-gather of shard sizes, the float64
MAX all-reduce of the timing, all_gather_object, barrier."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    import probing_rag_amd as pra
    from oracle import oracle_np as onp
    N, d, k, B = 60_000, 768, 10, 64
    ix = pra.ShardedFlatIndex(d, "cos", "f16")
    ix.add_synthetic_local(42, 0, N)
    ix.sync()
    q = torch.from_numpy(onp.synth_rows(7, 0, B, d)).cuda()
    D0, I0 = ix.search(q, k)                                   # world 1: plain local search
    # the multi-rank data path, forced: packed local result -> ONE all_gather_into_tensor -> packed merge
    buf, D_loc, I_loc, _ = ix.engine.search_packed(q, k, ix.id_offset, 1)
    assert buf.dtype == torch.uint8 and buf.dim() == 2 and buf.shape[0] == 1
    gathered = torch.empty((1, buf.shape[1]), dtype=torch.uint8, device=buf.device)    # [world, stride]
    dist.all_gather_into_tensor(gathered, buf)
    D1, I1 = ix.engine.merge_packed(gathered, B, k, ix.metric)
    assert torch.equal(I1, I0) and torch.equal(D1, D0)
    # the exchange in C (include/prag.h: prag_rccl_* + prag_index_set_comm + prag_index_search_sharded): a
    # communicator of the library's own over the one rank, then the whole sharded search as ONE call - local search
    # with tagged ids, ncclAllGather on the current stream, merge
    assert ix._comm is None and ix.exchange == "none"          # opt-in (PRAG_C_EXCHANGE=1), never by itself
    ok, why = ix.enable_c_exchange()
    assert ok and ix.exchange == "prag_rccl", "RCCL communicator of the library's own could not be created: " + why
    ix.engine.index.profile(16)                                # the exchange step has an event ring of its own
    D2, I2 = ix.search(q, k)
    torch.cuda.synchronize()
    xms = ix.engine.index.profile_read_exchange()
    assert len(xms) == 1 and 0.0 < xms[0] < 50.0, xms
    ix.engine.index.profile(0)
    assert torch.equal(I2, I0) and torch.equal(D2, D0)
    for B2 in (1, 300):                                        # the reference's call shape, and the tiled scans
        q2 = torch.from_numpy(onp.synth_rows(8, 0, B2, d)).cuda()
        Da, Ia = ix.engine.index.search(q2, k)
        Db, Ib = ix.search(q2, k)
        assert torch.equal(Ia, Ib) and torch.equal(Da, Db)
    # ... captured into a graph and replayed: nothing in the call waits for the stream
    out = (torch.empty_like(D0), torch.empty_like(I0))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ix.engine.index.search_sharded(q, k, ix.id_offset, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ix.engine.index.search_sharded(q, k, ix.id_offset, out=out)
    out[1].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[1], I0) and torch.equal(out[0], D0)
    del g
    ix.disable_c_exchange()
    assert ix._comm is None and ix.exchange == "none"
    # PRAG_C_EXCHANGE=1 at construction: the C exchange or an exception, never a silent fallback
    os.environ["PRAG_C_EXCHANGE"] = "1"
    ix2 = pra.ShardedFlatIndex(d, "cos", "f16")
    assert ix2.exchange == "prag_rccl" and ix2._comm is not None
    ix2.add_synthetic_local(42, 0, 5000)
    ix2.sync()
    D3, I3 = ix2.search(q, k)
    Dl, Il = ix2.engine.index.search(q, k)
    assert torch.equal(I3, Il) and torch.equal(D3, Dl)
    ix2.close()
    del os.environ["PRAG_C_EXCHANGE"]
    # shard sizes (ShardedFlatIndex.sync), timing reduction and device list (bench.py)
    counts = torch.zeros(1, dtype=torch.int64, device="cuda")
    dist.all_gather_into_tensor(counts, torch.tensor([N], dtype=torch.int64, device="cuda"))
    assert counts.tolist() == [N]
    t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    devs = [None]
    dist.all_gather_object(devs, 0)
    assert devs == [0]
    dist.barrier()
    torch.cuda.synchronize()
    print("RCCL_ONE_RANK_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
