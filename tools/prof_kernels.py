#!/usr/bin/env python3
"""Tiny driver for rocprofv3 runs: launches the two hot kernels a few times on
the BASELINE workload shapes (no timing, no CPU work).

  rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_kernels.py
  rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/prof_kernels.py --docs 21000000
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=21_000_000)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--gate-batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--store", default="f16")
    ap.add_argument("--metric", default="cos")
    ap.add_argument("--skip-gate", action="store_true")
    ap.add_argument("--shadow", type=int, default=1)
    ap.add_argument("--skip-scan", action="store_true")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--weights", default="f16", choices=["f16", "f32"], help="prober weight mode of the gate launches")
    ap.add_argument("--x-dtype", default="f16", choices=["f16", "f32"], help="element type of the pooled states")
    args = ap.parse_args()
    import torch
    import probing_rag_amd as pra
    from probing_rag_amd.synth import random_prober_state, synth_rows
    torch.cuda.set_device(0)
    if not args.skip_scan:
        ix = pra.HipFlatIndex(768, args.metric, args.store, capacity=args.docs)
        ix.add_synthetic(42, 0, args.docs)
        ix.set_shadow(args.shadow)
        q = torch.from_numpy(synth_rows(7, 0, args.queries, 768)).cuda()
        for _ in range(args.iters):
            ix.search(q, args.k)
    if not args.skip_gate:
        ens = pra.HipProberEnsemble(6, 2048, 2, weights=args.weights)
        for l in range(6):
            ens.load_layer(l, random_prober_state(100 + l, 2048))
        x = torch.randn((6, args.gate_batch, 2048), device="cuda")
        if args.x_dtype == "f16":
            x = x.half()
        for _ in range(args.iters):
            ens.gate(x, 0, 0.0)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
