#!/usr/bin/env python3
"""One rank's pass of the 8-GPU job (gate over 512 pooled states, two-level top-10 of 64 queries over 2 625 000 rows) a
few hundred times: the program to put after `rocprofv3 --kernel-trace --stats --`; prints the host-timed pass itself."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
import bench
from probing_rag_amd.synth import synth_rows, random_prober_state
n = int(os.environ.get("SHARD_REPS", 200))
L, dm, d = bench.N_LAYERS, bench.D_MODEL, bench.D_EMB
ens = pra.HipProberEnsemble(L, dm, 2, weights="f16")
for l in range(L):
    ens.load_layer(l, random_prober_state(100 + l, dm))
ix = pra.HipFlatIndex(d, "cos", "f16", capacity=bench.SHARD_ROWS)
ix.add_synthetic(42, 0, bench.SHARD_ROWS)
ix.set_shadow(1)
ix.prepare()
if os.environ.get("SHARD_SCAN_WG"):
    ix.set_scan_workgroups(int(os.environ["SHARD_SCAN_WG"]))
q = torch.from_numpy(synth_rows(7, 0, bench.SHARD_QUERIES, d)).cuda()
g = torch.Generator(device="cuda").manual_seed(4321)
Bg = bench.SHARD_GATE_ROWS
x = torch.randn((L, Bg, dm), generator=g, device="cuda", dtype=torch.float32).half()
gate_out = (torch.empty((L, Bg, 2), dtype=torch.float32, device="cuda"), torch.empty((Bg, 2), dtype=torch.float32, device="cuda"),
            torch.empty((Bg,), dtype=torch.int32, device="cuda"))
out = (torch.empty((bench.SHARD_QUERIES, 10), dtype=torch.float32, device="cuda"),
       torch.empty((bench.SHARD_QUERIES, 10), dtype=torch.int64, device="cuda"))
side = torch.cuda.Stream()
TAIL = os.environ.get("SHARD_TAIL") == "1"      # the gate beside the search's tail (prag_index_stream_wait_scan)
if TAIL:
    ix.stream_wait_scan(side)
FUSED = os.environ.get("SHARD_FUSED") == "1"    # prag_search_and_gate: the gate in the bound kernel's launch
def one():
    if FUSED:
        pra.search_and_gate(ix, q, 10, ens, x, 0, 0.0, out=out, gate_out=gate_out)
        return
    if not TAIL:
        ens.gate(x, 0, 0.0, out=gate_out)
        ix.search(q, 10, out=out)
        return
    ix.search(q, 10, out=out)
    ix.stream_wait_scan(side)
    with torch.cuda.stream(side):
        ens.gate(x, 0, 0.0, out=gate_out)
    torch.cuda.current_stream().wait_stream(side)
for _ in range(12):
    one()
    torch.cuda.synchronize()
for _ in range(20):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    one()
torch.cuda.synchronize()
print(f"shard pass: {(time.perf_counter() - t0) / n * 1e3:.4f} ms; plan: {ix.last_plan()}", flush=True)
