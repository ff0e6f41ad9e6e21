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
