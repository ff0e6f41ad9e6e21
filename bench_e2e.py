"""`python bench.py --e2e`: BASELINE config 5 as a TIMED run - the retrieve-decide loop of exp_rag.py:396-474
around a Gemma-2B-SHAPED decoder (HF GemmaConfig 18 x 2048, random weights: the checkpoint is not in this
image) with forward hooks on layers 6..16 (exp_rag.py:311-329), twice over the same queries:

  hip        hooks -> HiddenStatePool (on-device running sum) -> fused 6-prober gate -> row-sharded HIP flat
             index (device queries in, ids out) -> passage lookup
  reference  the reference's own data path: every hook does `activations.detach().cpu()` (6 D2H syncs per
             token, exp_rag.py:317-321), `cat(cache[1:]).to(device).sum(1)` + six eager-PyTorch prober calls
             with `.to('cpu')` each (exp_rag.py:381-389, utils.py:389-390), softmax / sum / threshold in Python
             (exp_rag.py:406-415), and a CPU flat L2 scan (torch-cpu, bounded sub-corpus: faiss-cpu is not in
             this image and a 21 M x 768 float32 host scan per query is not run - the figure is a LOWER bound
             on the reference's retrieval time)

Reported: queries/s of both paths and the share of wall time in LM generation / gate / retrieval.
Token ids, query embeddings (contriever is absent) and passages are synthetic; the LM work is real-size.
Multi-rank (`--gpus N`): queries are split across ranks for generation and gating; a retrieval is a lockstep
collective - every rank contributes its pending query (or none), the gathered queries are searched on every
shard, local top-k all-gathered (RCCL) and merged, each rank keeps its row.
"""
import time

import numpy as np

LAYERS = list(range(6, 17, 2))          # exp_rag.py:311
D_MODEL, D_EMB = 2048, 768


class _Clock:
    """wall time per category, GPU drained at the boundaries"""

    def __init__(self, torch):
        self.torch = torch
        self.t = {"generate": 0.0, "gate": 0.0, "retrieve": 0.0}
        self.n = {"generate": 0, "gate": 0, "retrieve": 0}

    def run(self, kind, fn):
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        self.torch.cuda.synchronize()
        self.t[kind] += time.perf_counter() - t0
        self.n[kind] += 1
        return out


def _clock_overhead_us(torch, n=200):
    c = _Clock(torch)
    for _ in range(n):
        c.run("gate", lambda: None)
    return c.t["gate"] / n * 1e6


class _Lockstep:
    """One retrieval step across all ranks: gather the pending queries, search every shard, merge."""

    def __init__(self, torch, dist, index, world, rank, dev, k):
        self.torch, self.dist, self.index, self.world, self.rank, self.dev, self.k = torch, dist, index, world, rank, dev, k
        self.q_all = torch.zeros((world, D_EMB), dtype=torch.float32, device=dev)
        self.flags = torch.zeros((world, 2), dtype=torch.float32, device=dev)      # (has a query, has finished everything)

    def step(self, q, finished=False):
        """q: [1,768] device tensor or None.  Returns (ids of this rank's query or None, every rank finished)."""
        torch = self.torch
        if self.world == 1:
            if q is None:
                return None, finished
            _, I = self.index.search(q, self.k)
            return I[0], finished
        mine_q = q if q is not None else torch.zeros((1, D_EMB), dtype=torch.float32, device=self.dev)
        mine_f = torch.tensor([[1.0 if q is not None else 0.0, 1.0 if finished else 0.0]], device=self.dev)
        self.dist.all_gather_into_tensor(self.q_all, mine_q.contiguous())
        self.dist.all_gather_into_tensor(self.flags, mine_f)
        flags = self.flags.cpu()
        if float(flags[:, 0].sum()) > 0:
            _, I = self.index.search(self.q_all, self.k)       # replicated queries -> identical merged result
            ids = I[self.rank] if q is not None else None
        else:
            ids = None
        return ids, bool(float(flags[:, 1].sum()) == self.world)


def _make_lm(torch, dev):
    from transformers import GemmaConfig, GemmaForCausalLM
    torch.manual_seed(0)
    cfg = GemmaConfig(vocab_size=256000, hidden_size=2048, intermediate_size=16384, num_hidden_layers=18,
                      num_attention_heads=8, num_key_value_heads=1, head_dim=256, max_position_embeddings=8192)
    with torch.device(dev):
        lm = GemmaForCausalLM(cfg)
    return lm.half().eval()


def _run_path(kind, torch, pra, lm, states, lock, my_queries, args, dev, cpu_sub=None):
    """One pass over this rank's queries with the `hip` or the `reference` data path."""
    clock = _Clock(torch)
    handles = []
    if kind == "hip":
        pool = pra.HiddenStatePool(len(LAYERS), D_MODEL, defer=True)    # HF decoder-layer outputs are fresh tensors
        ens = pra.HipProberEnsemble(len(LAYERS), D_MODEL, 2, weights="f32")
        for slot, st in enumerate(states):
            ens.load_layer(slot, st)
        for slot, l in enumerate(LAYERS):
            handles.append(lm.model.layers[l].register_forward_hook(
                lambda mod, inp, out, slot=slot: pool.observe(slot, out[0] if isinstance(out, tuple) else out)))
        reset = pool.reset
        if getattr(args, "e2e_step_gate", 1):
            # round 6: every decode step's pooling launch also runs the gate on the sums so far (prag_pool_step_gate);
            # the decision of the last step is in host memory when `generate` returns
            pool.attach_gate(ens, 0, args.e2e_theta)

            def gate():
                return int(pool.decide()[0])
        else:
            def gate():
                # one C call: gate kernels, the decision in host memory, the wait (prag_gate_decide)
                return int(ens.decide(pool.pooled(), ablation=0, threshold=args.e2e_theta)[0])
    else:
        from oracle import torch_cpu                     # baseline leg: the reference-shaped eager modules
        probers = [m.to(dev) for m in torch_cpu.make_probers(states, D_MODEL)]
        cache = {}
        for slot, l in enumerate(LAYERS):
            def hook_fn(mod, inp, out, slot=slot):      # exp_rag.py:317-321
                act = out[0] if isinstance(out, tuple) else out
                cache.setdefault(slot, []).append(act.detach().cpu())
            handles.append(lm.model.layers[l].register_forward_hook(hook_fn))
        reset = cache.clear
        softmax_f = torch.nn.Softmax(dim=1)

        def gate():
            with torch.no_grad():
                logits = []
                for slot, p in enumerate(probers):      # exp_rag.py:381-389, utils.py:389-390
                    x = torch.cat(cache[slot][1:], dim=1).to(dev)
                    logits.append(p(torch.sum(x, dim=1).float()).to("cpu"))
            s = softmax_f(logits[0])
            for n in range(1, len(logits)):             # exp_rag.py:406-415
                s = s + softmax_f(logits[n])
            s = s.squeeze()
            return 0 if float(s[0]) + args.e2e_theta < float(s[1]) else 1

    rng = np.random.default_rng(1234)
    counts, finished_all = [], False
    from probing_rag_amd.synth import synth_rows
    for qi in my_queries:
        prompt = torch.from_numpy(rng.integers(5, 250000, size=(1, args.e2e_prompt_len))).to(dev)
        state = {"round": 0}

        def generate(ids):
            return clock.run("generate", lambda: lm.generate(ids, max_new_tokens=args.e2e_new_tokens, do_sample=False,
                                                              use_cache=True, pad_token_id=0))

        def retrieve(text, k):
            emb = synth_rows(900 + qi, state["round"], 1, D_EMB)        # stands in for contriever.encode(text)
            state["round"] += 1
            if kind == "hip":
                ids, _ = clock.run("retrieve", lambda: lock.step(torch.from_numpy(emb).to(dev)))
                return None, ids.unsqueeze(0)
            xs, xn = cpu_sub
            _, I = clock.run("retrieve", lambda: torch_cpu.flat_search(xs, xn, torch.from_numpy(emb), k, True))
            return None, I

        pred, rc = pra.retrieve_decide(
            f"question {qi}", prompt, generate=generate, gate=lambda: clock.run("gate", gate), retrieve=retrieve,
            lookup=lambda ids: [f"passage {i}" for i in ids],
            make_prompt=lambda q, evid: evid,
            tokenize=lambda s: torch.cat([prompt, torch.from_numpy(rng.integers(5, 250000, size=(1, 5 * 100))).to(dev)], 1),
            to_string=lambda out: ["decoded text"], reset=reset, k=5)
        counts.append(rc)
    if kind == "hip":                       # keep serving the other ranks' retrievals until everyone is done
        while not finished_all:
            _, finished_all = lock.step(None, finished=True)
    for h in handles:
        h.remove()
    return clock, counts


def run(args, pra, torch, dist, world, rank, dev_index):
    dev = torch.device("cuda", dev_index)
    from probing_rag_amd.synth import random_prober_state, synth_rows
    lo, hi = pra.partition_rows(args.docs, world, rank)
    index = pra.ShardedFlatIndex(D_EMB, "l2", args.store, capacity=hi - lo)     # the reference's IndexFlatL2 (make_indexer.py:450)
    index.add_synthetic_local(42, lo, hi - lo)
    index.sync()
    lm = _make_lm(torch, dev)
    states = [random_prober_state(100 + l, D_MODEL) for l in range(len(LAYERS))]
    my_queries = list(range(rank, args.e2e_queries, world))
    lock = _Lockstep(torch, dist, index, world, rank, dev, 5)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: one query through the HIP path (kernel loads, allocator, generate's caches)
    wa = argparse_copy(args, e2e_new_tokens=4)
    _run_path("hip", torch, pra, lm, states, _Lockstep(torch, dist, index, world, rank, dev, 5), my_queries[:1], wa, dev)
    fence()
    t0 = time.perf_counter()
    clock, counts = _run_path("hip", torch, pra, lm, states, lock, my_queries, args, dev)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = {
        "metric": "end-to-end queries/sec (retrieve-decide loop, Gemma-2B-shaped LM + gate + sharded retriever)",
        "value": args.e2e_queries / dt, "unit": "queries/s", "n_gpus": world, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "LM fp16 (PyTorch-ROCm eager); gate and index as in the default line",
        "data": "synthetic (random-weight Gemma-2B-shaped decoder, synthetic token ids / query embeddings / passages)",
        "config": {"workload": f"BASELINE config 5 shape: {args.e2e_queries} queries, <= 4 retrieval rounds each, "
                               f"{args.e2e_prompt_len}(+500) prompt tokens, {args.e2e_new_tokens} new tokens per generation, "
                               f"IndexFlatL2 over {args.docs} x 768 {args.store} docs row-sharded x{world}, k=5",
                   "timed_region_s": dt, "queries": args.e2e_queries, "queries_this_rank": len(my_queries)},
        "hip_path": {"wall_s_rank0": sum(clock.t.values()), "seconds": clock.t, "calls": clock.n,
                     # every timed call is bracketed by two device synchronisations: their own cost, measured on an
                     # empty call, is in every per-call figure (the gate's ~30 us - bench.py gate_b1_latency - sits under it)
                     "clock_overhead_us_per_call": _clock_overhead_us(torch),
                     "us_per_call": {k_: (clock.t[k_] / clock.n[k_] * 1e6 if clock.n[k_] else None) for k_ in clock.t},
                     "share": {k: v / max(1e-12, sum(clock.t.values())) for k, v in clock.t.items()},
                     "retrieval_rounds_histogram": {str(c): counts.count(c) for c in sorted(set(counts))}},
    }
    if rank == 0 and not args.no_cpu_baseline:
        # the reference's data path on the same queries (this rank's), same LM, same decisions up to rounding
        n_sub = min(args.docs, args.e2e_cpu_docs)
        xs = torch.from_numpy(synth_rows(42, 0, n_sub, D_EMB))
        cpu_sub = (xs, (xs * xs).sum(1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rclock, rcounts = _run_path("reference", torch, pra, lm, states, None, my_queries, args, dev, cpu_sub)
        torch.cuda.synchronize()
        rdt = time.perf_counter() - t0
        out["reference_style_path"] = {
            "queries_per_s_this_rank": len(my_queries) / rdt, "wall_s": rdt, "seconds": rclock.t, "calls": rclock.n,
            "share": {k: v / max(1e-12, sum(rclock.t.values())) for k, v in rclock.t.items()},
            "retrieval_rounds_histogram": {str(c): rcounts.count(c) for c in sorted(set(rcounts))},
            "same_rounds_as_the_hip_path": rcounts == counts,
            "what": "hooks with activations.detach().cpu() per token and layer, cat/sum + six eager prober calls with "
                    ".to('cpu'), Python gate; retrieval = torch-cpu flat L2 over a "
                    f"{n_sub}-row sub-corpus (lower bound: faiss-cpu absent, the full {args.docs}-row host scan is not run)",
            "hip_over_reference_queries_per_s_this_rank": (len(my_queries) / max(1e-12, sum(clock.t.values()))) / (len(my_queries) / rdt)}
    return out


def gate_in_loop(torch, pra, states, dev, gens=24, new_tokens=16, prompt_len=64, lm=None, modes=("step", "decide")):
    """The loop's gate call by itself (the back-to-back figure is not what the loop sees): a
    Gemma-2B-shaped decoder generates `new_tokens` tokens (hooks -> HiddenStatePool), then the gate is timed exactly as
    the loop's clock does - device drained on both sides - once with the decode steps' launches deciding as they go
    (`step`: HiddenStatePool.attach_gate, prag_pool_step_gate) and once with round 5's call after `generate` has
    returned (`decide`: prag_gate_decide).  Median us per gate call and ms per generate call, per mode."""
    lm = lm or _make_lm(torch, dev)
    out = {"generations_per_mode": gens, "new_tokens": new_tokens}
    rng = np.random.default_rng(0)
    for mode in modes:
        pool = pra.HiddenStatePool(len(LAYERS), D_MODEL, defer=True)
        ens = pra.HipProberEnsemble(len(LAYERS), D_MODEL, 2, weights="f32")
        for slot, st in enumerate(states):
            ens.load_layer(slot, st)
        if mode == "step":
            pool.attach_gate(ens, 0, 0.0)
        handles = [lm.model.layers[l].register_forward_hook(
            lambda mod, inp, o, slot=slot: pool.observe(slot, o[0] if isinstance(o, tuple) else o))
            for slot, l in enumerate(LAYERS)]
        t_gen, t_gate = [], []
        for g in range(gens + 3):
            pool.reset()
            prompt = torch.from_numpy(rng.integers(5, 250000, size=(1, prompt_len))).to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lm.generate(prompt, max_new_tokens=new_tokens, do_sample=False, use_cache=True, pad_token_id=0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            d = int(pool.decide()[0]) if mode == "step" else int(ens.decide(pool.pooled(), 0, 0.0)[0])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if g >= 3:
                t_gen.append(t1 - t0)
                t_gate.append(t2 - t1)
        for h in handles:
            h.remove()
        ens.close()
        out[mode + "_gate_us"] = float(np.median(t_gate)) * 1e6
        out[mode + "_gate_us_p90"] = float(np.percentile(t_gate, 90)) * 1e6
        out[mode + "_generate_ms"] = float(np.median(t_gen)) * 1e3
    return out


def gate_cold(torch, pra, states, dev, reps=60, flush_bytes=512 << 20):
    """ens.decide at B = 1 with a 512 MB read between calls (weights, kernel code and workspaces out of L2 / MALL; the
    host stays warm - that part only the loop shows, gate_in_loop above): median us per call, every call waited for."""
    ens = pra.HipProberEnsemble(len(LAYERS), D_MODEL, 2, weights="f32")
    for slot, st in enumerate(states):
        ens.load_layer(slot, st)
    x = torch.randn((len(LAYERS), 1, D_MODEL), device=dev)
    flush = torch.empty(flush_bytes, dtype=torch.uint8, device=dev)
    ts = []
    for i in range(reps + 5):
        flush.view(torch.int32).sum()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ens.decide(x, 0, 0.0)
        t1 = time.perf_counter()
        if i >= 5:
            ts.append(t1 - t0)
    ens.close()
    del flush
    return float(np.median(ts)) * 1e6


def argparse_copy(args, **kw):
    import copy
    a = copy.copy(args)
    for k, v in kw.items():
        setattr(a, k, v)
    return a
