#!/usr/bin/env python3
"""Where the in-loop gate call spends its time (97 us per call inside the retrieve-decide loop
against 30 us back to back).  The loop of bench_e2e.py in miniature: a Gemma-2B-shaped decoder generates N tokens
(hooks -> HiddenStatePool), then the gate is timed in pieces - pooled() (the deferred flush), decide() (host call
until the decision is in host memory) - with the device drained on both sides, as bench_e2e's clock does.

  python tools/gate_in_loop.py [--gens 40] [--new-tokens 16] [--mode decide|speculate] [--pollute lm|flush|none]

`--pollute flush` replaces the LM by a 1 GiB read (what bench.py's cold figure does), `none` is back to back.
Put the program after `rocprofv3 --kernel-trace --` and feed the trace to tools/gate_trace_gaps.py for the kernels'
own durations and the gaps between them."""
import argparse
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench_e2e
import probing_rag_amd as pra
from probing_rag_amd.synth import random_prober_state

ap = argparse.ArgumentParser()
ap.add_argument("--gens", type=int, default=40)
ap.add_argument("--new-tokens", type=int, default=16)
ap.add_argument("--prompt-len", type=int, default=64)
ap.add_argument("--mode", default="decide")
ap.add_argument("--pollute", default="lm", choices=["lm", "flush", "none"])
ap.add_argument("--weights", default="f32")
ap.add_argument("--probe-first", type=int, default=0,
                help="1: before the gate, time one trivial launch + wait (a 4-byte add): what ANY first launch after "
                     "`generate` costs on this host")
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L, D = len(bench_e2e.LAYERS), bench_e2e.D_MODEL
ens = pra.HipProberEnsemble(L, D, 2, weights=args.weights)
for l in range(L):
    ens.load_layer(l, random_prober_state(100 + l, D))
pool = pra.HiddenStatePool(L, D, defer=True)
if args.mode == "speculate":
    pool.attach_gate(ens, 0, 0.0)
lm = None
if args.pollute == "lm":
    lm = bench_e2e._make_lm(torch, dev)
    for slot, l in enumerate(bench_e2e.LAYERS):
        lm.model.layers[l].register_forward_hook(
            lambda mod, inp, out, slot=slot: pool.observe(slot, out[0] if isinstance(out, tuple) else out))
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
rng = np.random.default_rng(0)
xs = torch.randn((L, 1, D), device=dev)

t_gen, t_pool, t_dec, t_probe = [], [], [], []
tiny = torch.zeros(4, device=dev)
decisions = []
for g in range(args.gens + 3):
    pool.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if lm is not None:
        prompt = torch.from_numpy(rng.integers(5, 250000, size=(1, args.prompt_len))).to(dev)
        lm.generate(prompt, max_new_tokens=args.new_tokens, do_sample=False, use_cache=True, pad_token_id=0)
    else:
        for slot in range(L):
            pool.observe(slot, xs[slot:slot + 1])           # the prompt pass (skipped by the pool)
        for _ in range(3):
            for slot in range(L):
                pool.observe(slot, xs[slot:slot + 1])
        if args.pollute == "flush":
            flush.view(torch.int32).sum()
    torch.cuda.synchronize()
    if args.probe_first:
        tp = time.perf_counter()
        tiny.add_(1.0)
        torch.cuda.synchronize()
        if g >= 3:
            t_probe.append(time.perf_counter() - tp)
    t1 = time.perf_counter()
    acc = pool.pooled()
    t2 = time.perf_counter()
    if args.mode == "speculate":
        d = int(pool.decide()[0])
    else:
        d = int(ens.decide(acc, 0, 0.0)[0])
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    if g >= 3:
        t_gen.append(t1 - t0)
        t_pool.append(t2 - t1)
        t_dec.append(t3 - t2)
        decisions.append(d)


def med(v):
    return float(np.median(v)) * 1e6


print(f"mode={args.mode} pollute={args.pollute} weights={args.weights} gens={args.gens} new_tokens={args.new_tokens}: "
      f"generate {med(t_gen) / 1e3:.2f} ms | pooled() {med(t_pool):.1f} us | decide {med(t_dec):.1f} us "
      f"(p10 {np.percentile(t_dec, 10) * 1e6:.1f}, p90 {np.percentile(t_dec, 90) * 1e6:.1f}) | "
      f"gate total {med(np.add(t_pool, t_dec)):.1f} us | retrieve rate {np.mean(decisions):.2f}"
      + (f" | trivial launch + wait before the gate {med(t_probe):.1f} us" if t_probe else ""), flush=True)
