#!/usr/bin/env python3
"""Benchmark of the Probing-RAG retrieval-gating hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input:
  gate   : fused 6-layer prober ensemble + softmax/sum/threshold over B_gate
           pooled hidden states (d_model 2048, fp16)           [exp_rag.py:406-415]
  search : cosine top-10 of B_q query embeddings over the row-sharded
           N_docs x 768 fp16 corpus (local fused scan/top-k, all-gather of the
           local top-k over RCCL, (score,id) merge)             [utils.py:378-380]
Workload = BASELINE.json's quoted sizes: d_model=2048, N_docs=21M, d_emb=768.
Total work is fixed as GPUs are added ("strong"): the 21M rows and the 4096
gate rows are split across ranks.

  python bench.py                      # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N

Prints ONE JSON line on rank 0 (contract in the task statement) with extra
objects `roofline` (dominant kernel: scan_topk, HBM-bound) and `cpu_baseline`.
Other sizes are parity/diagnostic cases, e.g. BASELINE config 3:
  python bench.py --queries 1000 --docs 1000000      # MFMA-tiled scan, roofline.bound = "mfma"
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F16_PEAK_TF = 2500.0  # dense fp16/bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=21_000_000, help="total corpus rows (all GPUs)")
    ap.add_argument("--queries", type=int, default=64, help="query embeddings per step")
    ap.add_argument("--gate-batch", type=int, default=4096, help="pooled hidden states per step (all GPUs)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="cos", choices=["cos", "l2", "ip"])
    ap.add_argument("--store", default="f16", choices=["f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-gate", type=int, default=1,
                    help="1: run the gate on a second stream beside the HBM-bound scan (scan capped at "
                         "n_cu-16 workgroups); 0: one stream")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(args, d_emb, states):
    """Rank 0, N=1 only: the torch-cpu restatement (oracle/torch_cpu.py) timed on
    this box's host cores over a bounded sample of the same workload."""
    import torch
    from oracle import oracle_np as onp, torch_cpu
    cores = torch.get_num_threads()
    # ---- scoring sample: B_q queries x Ns docs, same metric/k ------------------
    Bq, k = args.queries, args.k
    Ns = 200_000
    xs = torch.from_numpy(onp.synth_rows(42, 0, Ns, d_emb))
    if args.metric == "cos":
        xs = torch.nn.functional.normalize(xs, dim=1)
    xn = (xs * xs).sum(1)
    q = torch.from_numpy(onp.synth_rows(7, 0, Bq, d_emb))
    torch_cpu.flat_search(xs, xn, q, k, args.metric == "l2")          # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        torch_cpu.flat_search(xs, xn, q, k, args.metric == "l2")
        reps += 1
        if time.perf_counter() - t0 > args.cpu_seconds * 0.6 or reps >= 200:
            break
    dt = time.perf_counter() - t0
    scores_per_s = Bq * Ns * reps / dt
    # ---- gate sample: BASELINE config 1 (128 states x 6 probers) --------------
    probers = torch_cpu.make_probers(states, 2048)
    x = torch.from_numpy(onp.synth_rows(1234, 0, 6 * 128, 2048).reshape(6, 128, 2048))
    torch_cpu.gate(probers, x)
    t0 = time.perf_counter()
    greps = 0
    while True:
        torch_cpu.gate(probers, x)
        greps += 1
        if time.perf_counter() - t0 > args.cpu_seconds * 0.3 or greps >= 400:
            break
    gdt = time.perf_counter() - t0
    return {
        "value": scores_per_s, "unit": "query*doc scores/s", "cores": cores, "kind": "port",
        "sample": f"torch-cpu flat {args.metric} top-{k}: {Bq} queries x {Ns} docs x {d_emb} fp32, "
                  f"{reps} reps in {dt:.1f}s (faiss-cpu not installed in this image)",
        "gate_decisions_per_s": 128 * greps / gdt,
        "gate_sample": f"torch-cpu 6 x ImprovedProbe(2048) + gate, B=128 fp32, {greps} reps in {gdt:.1f}s",
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(1, n_dev)   # (several ranks per GPU only in the gloo smoke mode below)
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") is the product path.  PRAG_BENCH_BACKEND=gloo exists only to exercise the
        # multi-rank control flow on a box with fewer GPUs than ranks (collectives staged via host).
        backend = os.environ.get("PRAG_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    import probing_rag_amd as pra
    from oracle import oracle_np as onp
    from tests.golden import cases

    d_model, d_emb, L = 2048, 768, 6
    # ---- corpus shard (generated on device, bit-identical to oracle_np.synth_rows)
    lo, hi = pra.partition_rows(args.docs, world, rank)
    index = pra.ShardedFlatIndex(d_emb, args.metric, args.store, capacity=hi - lo)
    index.add_synthetic_local(42, lo, hi - lo)
    index.sync()
    # ---- gate: 6 probers, this rank's slice of the pooled hidden states
    states = [cases.synth_state(100 + l, d_model) for l in range(L)]
    ens = pra.HipProberEnsemble(L, d_model, 2, weights="f16")
    for l, st in enumerate(states):
        ens.load_layer(l, st)
    glo, ghi = pra.partition_rows(args.gate_batch, world, rank)
    Bg = max(1, ghi - glo)
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    x = torch.randn((L, Bg, d_model), generator=g, device="cuda", dtype=torch.float32).half()
    gate_out = (torch.empty((L, Bg, 2), dtype=torch.float32, device="cuda"),
                torch.empty((Bg, 2), dtype=torch.float32, device="cuda"),
                torch.empty((Bg,), dtype=torch.int32, device="cuda"))
    # ---- queries: replicated on every rank; a few are planted near known rows
    q_np = onp.synth_rows(7, 0, args.queries, d_emb)
    n_plant = min(8, args.queries)
    planted = [(i * 2_654_435 + 17) % args.docs for i in range(n_plant)]
    for i, r in enumerate(planted):
        q_np[i] = onp.synth_rows(42, r, 1, d_emb)[0] + 0.05 * q_np[i]
    q = torch.from_numpy(q_np).cuda()

    main_stream = torch.cuda.current_stream()
    side_stream = torch.cuda.Stream()
    if args.overlap_gate:
        n_cu = torch.cuda.get_device_properties(dev_index).multi_processor_count
        index.engine.index.set_scan_workgroups(n_cu - 16)

    def step():
        if not args.overlap_gate:
            ens.gate(x, 0, 0.0, out=gate_out)
            return index.search(q, args.k)
        # the search goes first so the scan's persistent workgroups settle on their CUs; the gate
        # (independent work: the decisions of the NEXT batch) fills the CUs left free
        start = torch.cuda.Event()
        start.record(main_stream)                 # everything before this step
        out = index.search(q, args.k)
        side_stream.wait_event(start)             # NOT the search just enqueued
        with torch.cuda.stream(side_stream):
            ens.gate(x, 0, 0.0, out=gate_out)
        main_stream.wait_stream(side_stream)      # the step ends when both are done
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    slots = args.steps * (1 + (args.queries - 1) // 64) + 8
    index.engine.index.profile(slots)
    ens.profile(args.steps + 8)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        D, I = step()
    fence()
    dt = time.perf_counter() - t0
    scan_ms = index.engine.index.profile_read()
    gate_ms = ens.profile_read()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # ---- correctness riders (outside the timed region) -------------------------
    I_host = I.cpu().numpy()
    planted_ok = all(int(I_host[i, 0]) == planted[i] for i in range(n_plant))
    recall = None
    if rank == 0:
        # top-k recall against the oracle (exact float64 brute force, C) on a bounded
        # sub-corpus searched by the same kernels: first 200k rows, 16 of the queries
        from oracle import oracle_c
        n_sub = min(200_000, args.docs)
        sub = pra.HipFlatIndex(d_emb, args.metric, args.store, capacity=n_sub)
        sub.add_synthetic(42, 0, n_sub)
        qs = q[: min(16, args.queries)]
        _, I_sub = sub.search(qs, args.k)
        metric_id = {"l2": onp.METRIC_L2, "ip": onp.METRIC_IP, "cos": onp.METRIC_COS}[args.metric]
        _, I_ref = oracle_c.flat_search(sub.reconstruct_n(0, n_sub), qs.cpu().numpy(), args.k, metric_id)
        I_sub = I_sub.cpu().numpy()
        recall = float(np.mean([len(set(a) & set(b)) / args.k for a, b in zip(I_sub.tolist(), I_ref.tolist())]))
        exact_order = bool(np.array_equal(I_sub, I_ref))
        del sub

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    scores = args.queries * args.docs
    value = scores / (dt / args.steps)
    n_local = hi - lo
    elt = 2 if args.store == "f16" else 4
    alg_bytes = n_local * d_emb * elt + (n_local * 4 if args.metric == "l2" else 0)
    # which scan kernel served the step (mirrors prag_index_search's choice)
    tiled = args.queries > 128 and args.store == "f16"           # MFMA-tiled scan, 256-query tiles
    per_pass = 128 if (args.queries > 64 and args.store == "f16") else 64
    passes = 1 if tiled else 1 + (args.queries - 1) // per_pass
    scan_avg_ms = float(np.mean(scan_ms)) if scan_ms else float("nan")
    achieved = alg_bytes / (scan_avg_ms * 1e-3) / 1e9
    if tiled:
        # the profiled launch is the last corpus segment (segments: 2048 rows, then x16)
        seg0 = 2048
        while seg0 * 16 < n_local:
            seg0 *= 16
        rows_last = n_local - seg0 if n_local > 2048 else n_local
        mm_flops = 2.0 * args.queries * rows_last * d_emb
        mm_tf = mm_flops / (scan_avg_ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_scan_topk.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get("rows_per_launch") == n_local and rec.get("store") == args.store:
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    gate_avg_ms = float(np.mean(gate_ms)) if gate_ms else float("nan")
    gate_flops = 2.0 * L * (d_model * 512 + 512 * 512 + 512 * 2) * Bg
    out = {
        "metric": "probe-decisions/sec + query*doc scores/sec/GPU (value = query*doc scores/sec, whole job)",
        "value": value, "unit": "query*doc scores/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16 (MFMA, f32 accumulate; f64 rerank)", "data": "synthetic",
        "config": {"workload": f"gate B={args.gate_batch} x 6 layers x d_model=2048 fp16 + flat {args.metric} "
                               f"top-{args.k} of {args.queries} queries over {args.docs} x 768 {args.store} docs",
                   "docs_total": args.docs, "docs_per_gpu": n_local, "d_emb": d_emb, "queries": args.queries,
                   "k": args.k, "gate_batch": args.gate_batch, "gate_batch_per_gpu": Bg, "d_model": d_model,
                   "parallelism": f"corpus rows sharded x{world}; gate rows split x{world}",
                   "gate_overlapped_with_scan": bool(args.overlap_gate)},
        "probe_decisions_per_s": args.gate_batch / (gate_avg_ms * 1e-3) if gate_ms else None,
        "scores_per_s_per_gpu": value / world,
        "planted_top1_recall": 1.0 if planted_ok else 0.0,
        "recall_at_k_vs_oracle": recall, "topk_ids_bit_exact_vs_oracle": exact_order,
        "recall_sample": f"{min(16, args.queries)} queries x {min(200_000, args.docs)} docs, float64 C oracle",
        "roofline": ({"bound": "mfma", "kernel": "scan_mm_kernel", "achieved": mm_tf, "peak": MFMA_F16_PEAK_TF,
                      "unit": "TFLOP/s", "frac": mm_tf / MFMA_F16_PEAK_TF, "traffic": None,
                      "algorithmic_flops_per_launch": mm_flops, "rows_in_launch": rows_last,
                      "avg_launch_ms": scan_avg_ms, "launches_per_step": passes} if tiled else
                     {"bound": "hbm", "kernel": "scan_qs_kernel" if per_pass == 128 else "scan_topk_kernel",
                      "achieved": achieved, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                      "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": scan_avg_ms,
                      "launches_per_step": passes}),
        "roofline_gate": {"bound": "mfma", "kernel": "prober_fused_kernel", "achieved": gate_flops / (gate_avg_ms * 1e-3) / 1e12,
                          "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                          "frac": gate_flops / (gate_avg_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF,
                          "avg_launch_ms": gate_avg_ms,
                          "hbm_GBs": (L * Bg * d_model * 2 + L * 1318914 * 2) / (gate_avg_ms * 1e-3) / 1e9},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, d_emb, states)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
