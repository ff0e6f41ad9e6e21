#!/usr/bin/env python3
"""Benchmark of the Probing-RAG retrieval-gating hot path on MI355X.

One "step" = `--inner` passes (default 20) of the hot path over one batch of synthetic input:
  gate   : fused 6-layer prober ensemble + softmax/sum/threshold over B_gate
           pooled hidden states (d_model 2048, fp16)           [exp_rag.py:406-415]
  search : cosine top-10 of B_q query embeddings over the row-sharded
           N_docs x 768 fp16 corpus (local two-level exact search: 8-bit shadow
           scan + proof-carrying filter + float64 rerank; --shadow 0 scans the
           fp16 rows directly; all-gather of the local top-k over RCCL,
           (score,id) merge)                                   [utils.py:378-380]
(`--inner` exists so that the timed region lasts seconds, not 0.1 s; `value` counts every pass.)
Workload = BASELINE.json's quoted sizes: d_model=2048, N_docs=21M, d_emb=768.
Total work is fixed as GPUs are added ("strong"): the 21M rows and the 4096
gate rows are split across ranks.

  python bench.py                      # 1 GPU
  python bench.py --gpus N             # starts its own torch.distributed.run child (one rank per GPU, RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N     # same thing

Rank 0's LAST stdout line is the compact JSON record (contract in the task statement, < 6 KB: compact_record); the
full record goes out one line earlier, prefixed `BENCH_DETAIL `, and into gpurun_out/bench_detail.json.  Extra objects:
  roofline      dominant kernel of the headline step (scan8 - or scan_topk with --shadow 0 -, HBM-bound),
                measured live with HIP events on the launch stream inside libprag; algorithmic bytes =
                the rows in the form that is scanned; `traffic` = HBM bytes per launch from two child
                `rocprofv3 --pmc` passes started by this run after the timed region (--measure-traffic 0, a run
                under a profiler, or a box without rocprofv3: the committed summary under profiles/ is quoted)
  variants      (1 GPU) the other call shapes SURVEY.md section 8d names, same corpus size:
                the reference's literal call (float32 rows, squared L2, k=5, one query) and fp16
                rows with 1 / 32 / 1000 queries - each with its own roofline fraction
  cpu_baseline  (1 GPU) BASELINE config 1 (128 states x 6 probers; 128 queries x 10k docs L2 top-5)
                and the reference's batch-1 call shape, torch-cpu and the C restatement, on this
                box's host cores
`python bench.py --e2e [--gpus N]` times BASELINE config 5 instead (bench_e2e.py): the retrieve-decide loop around a
Gemma-2B-shaped decoder, HIP path next to the reference's data path; its own JSON line, never the default.
Other sizes are parity/diagnostic cases, e.g. BASELINE config 3:
  python bench.py --queries 1000 --docs 1000000      # MFMA-tiled scan, roofline.bound = "mfma"
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F16_PEAK_TF = 2500.0  # dense fp16/bf16 MFMA
MFMA_I8_PEAK_TOP = 5000.0  # int8 MFMA: 2x the bf16 rate per clock (MI355X_MICROARCH.md, MFMA table)
MM8_MIN_ROWS = 2 << 20     # prag_index_search takes the int8 tiles (> 128 queries, shadow kept) from this shard size on
D_MODEL, D_EMB, N_LAYERS = 2048, 768, 6


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--inner", type=int, default=20, help="passes of the hot path per step")
    ap.add_argument("--docs", type=int, default=21_000_000, help="total corpus rows (all GPUs)")
    ap.add_argument("--queries", type=int, default=64, help="query embeddings per pass")
    ap.add_argument("--gate-batch", type=int, default=4096, help="pooled hidden states per pass (all GPUs)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="cos", choices=["cos", "l2", "ip"])
    ap.add_argument("--store", default="f16", choices=["f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--shadow", type=int, default=1,
                    help="1: two-level exact search through the 8-bit shadow of the rows (prag_index_set_shadow); "
                         "0: scan the stored rows themselves")
    ap.add_argument("--scan-workgroups", type=int, default=0,
                    help="workgroups of the corpus scan (0: the library's choice - 7/8 of the CUs for HBM-bound two-level scans)")
    ap.add_argument("--overlap-gate", type=int, default=3,
                    help="3 (default): one C call per pass and rank (prag_search_and_gate): the gate of the NEXT batch rides "
                         "in a launch of the local search - behind the scan's own workgroups when it fits under the scan "
                         "(the scan runs on 7/8 of the CUs; 21 M rows 2.65-2.68 -> 2.62-2.64 ms per pass, 8-GPU shard size "
                         "0.418 -> 0.408, profiles/r05u_scan_wg_sweep.txt), else beside the bound kernel - what follows the "
                         "corpus scan occupies a quarter of the chip (0.437 -> 0.415-0.418 ms per pass, 21 M rows 2.742 -> "
                         "2.696, profiles/r05j_*); 2: the gate of the NEXT batch on a second stream that waits for the search's corpus "
                         "scan only (prag_index_stream_wait_scan): it runs beside the search's tail - bound kernel, exact "
                         "rerank, fallback probes - measured 0.459 -> 0.454 ms per pass at the 8-GPU shard size, 0.420 "
                         "with the sampled pre-bound (profiles/r05c_shard_ab.txt); 0: one stream; 1: the gate on a second "
                         "stream beside the scan itself (scan capped at n_cu-16 workgroups; measured slower in rounds 3 "
                         "and 5: scan8 holds all of a CU's LDS)")
    ap.add_argument("--adaptive", type=int, default=0,
                    help="0 (default): the deterministic plan (prag_index_set_adaptive(ix, 0)) - every rank and every run "
                         "issues the same launches (scan on 7/8 of the CUs, nothing armed by history), the plan is in the "
                         "line; 1: the index times its scan grid on its first searches and arms tiers by history")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("PRAG_BENCH_LAUNCH_TIMEOUT", "1800")),
                    help="`--gpus N` without a launcher: seconds after which the torch.distributed.run child (its whole "
                         "process group) is killed and this process exits 124")
    ap.add_argument("--no-gate-in-loop", action="store_true",
                    help="skip the in-loop gate latency variant (builds a Gemma-2B-shaped random-weight decoder, ~15 s)")
    ap.add_argument("--no-shard-variant", action="store_true",
                    help="skip the 8-GPU shard shape (2 625 000 rows, 64 queries, 512 gate rows) in `variants`")
    ap.add_argument("--c-exchange-probe", type=int, default=int(os.environ.get("PRAG_BENCH_C_PROBE", "0")),
                    help="N > 1 under RCCL: after the timed region (which exchanges through torch.distributed unless "
                         "PRAG_C_EXCHANGE=1) create the library's own communicator, run the same passes through "
                         "prag_index_search_sharded and report ms per pass / all-gather us / identical ids as "
                         "`c_exchange_probe`.  A watchdog prints the line without it if the probe hangs.  Off by default: "
                         "the C exchange has never crossed an xGMI link (SCALE runs are the driver's)")
    ap.add_argument("--measure-traffic", type=int, default=1,
                    help="1 (default, 1 GPU): measure roofline.traffic in THIS run - two child `rocprofv3 --pmc` passes "
                         "(FETCH_SIZE, WRITE_SIZE; counters cannot be read inside the timed process) over "
                         "tools/prof_kernels.py at the run's shard size, after the timed region, ~10 s each at 21 M rows; "
                         "skipped when this process is itself being profiled or rocprofv3 is missing.  0: quote the "
                         "committed summary under profiles/ instead")
    # BASELINE config 5 as a timed run (not part of the default line): see bench_e2e.py
    ap.add_argument("--e2e", action="store_true",
                    help="time the retrieve-decide loop around a Gemma-2B-shaped random-weight decoder: HIP path vs the "
                         "reference's data path (hooks with .cpu(), eager probers, Python gate, CPU flat scan)")
    ap.add_argument("--e2e-queries", type=int, default=64)
    ap.add_argument("--e2e-new-tokens", type=int, default=16)
    ap.add_argument("--e2e-prompt-len", type=int, default=64)
    ap.add_argument("--e2e-theta", type=float, default=0.0)
    ap.add_argument("--e2e-step-gate", type=int, default=1,
                    help="1 (default): the pool's one launch per decode step also runs the gate on the sums so far "
                         "(HiddenStatePool.attach_gate / prag_pool_step_gate), the loop's gate call reads the decision from "
                         "host memory; 0: round 5's loop - prag_gate_decide after `generate` has returned")
    ap.add_argument("--e2e-cpu-docs", type=int, default=200_000,
                    help="rows of the sub-corpus the reference-style path scans on the CPU")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N=1): the ONLY part of this file that touches oracle/
# ---------------------------------------------------------------------------------------------
def _timed(fn, budget_s, max_reps=400):
    fn()                                   # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        fn()
        reps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or reps >= max_reps:
            return reps, dt


def _finite(o):
    """NaN / inf are not JSON: a figure that could not be measured is null in the record."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def cpu_baseline(args, states, q_c1, x_c1):
    """BASELINE config 1 - 128 synthetic hidden states (d=2048) through the 6-prober gate and a
    10k-doc flat L2 index, top-5 - timed on this box's host cores with (a) the torch-cpu port
    (the structure the reference runs: six module calls + softmax/sum/threshold; one sgemm + topk,
    faiss-cpu is not installable in this image) and (b) the plain-C restatement (prag_oracle.c,
    OpenMP), each also at the reference's real call shape (batch 1).  A bounded sample of the
    metric's own workload (B_q queries x 200k docs, same metric / k) is timed as well."""
    import torch
    from oracle import oracle_c, torch_cpu
    cores = torch.get_num_threads()
    slice_s = args.cpu_seconds / 8.0
    N1, k1 = 10_000, 5
    docs = torch.from_numpy(x_c1)                                  # [10k, 768] float32
    dn = (docs * docs).sum(1)
    q128 = torch.from_numpy(q_c1)                                  # [128, 768]
    out = {"cores": cores, "kind": "port", "unit": "query*doc scores/s"}
    # ---- flat L2 top-5, torch-cpu ---------------------------------------------------------
    r, dt = _timed(lambda: torch_cpu.flat_search(docs, dn, q128, k1, True), slice_s)
    out["value"] = 128 * N1 * r / dt
    out["sample"] = (f"config 1: torch-cpu flat L2 top-{k1}, 128 queries x {N1} docs x {D_EMB} fp32, {r} reps in "
                     f"{dt:.1f}s (faiss-cpu is not installed in this image)")
    out["sample_short"] = f"C1: torch-cpu flat L2 top-{k1}, 128 q x {N1} docs x {D_EMB} f32, {r} reps in {dt:.1f}s; no faiss here"
    r, dt = _timed(lambda: torch_cpu.flat_search(docs, dn, q128[:1], k1, True), slice_s, 4000)
    out["flat_b1_scores_per_s"] = N1 * r / dt
    out["flat_b1_ms_per_search"] = dt / r * 1e3
    # ---- gate, torch-cpu: thread sweep (all host threads on 512-wide GEMMs is slower than a few) -------
    probers = torch_cpu.make_probers(states, D_MODEL)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((N_LAYERS, 128, D_MODEL), generator=g)
    x1 = x[:, :1].contiguous()
    sweep = {}
    best = None
    for nt in sorted({min(8, cores), min(32, cores), cores}):
        torch.set_num_threads(nt)
        r, dt = _timed(lambda: torch_cpu.gate(probers, x), slice_s / 3.0)
        r1, dt1 = _timed(lambda: torch_cpu.gate(probers, x1), slice_s / 3.0, 4000)
        sweep[str(nt)] = {"b128_decisions_per_s": 128 * r / dt, "b1_decisions_per_s": r1 / dt1, "b1_ms": dt1 / r1 * 1e3}
        if best is None or 128 * r / dt > best[1]:
            best = (nt, 128 * r / dt, r, dt)
    torch.set_num_threads(cores)
    out["gate_decisions_per_s"] = best[1]
    out["gate_threads"] = best[0]
    out["gate_sample"] = (f"config 1: torch-cpu 6 x ImprovedProbe(2048) + gate, B=128 fp32, best of the thread sweep "
                          f"({best[0]} threads): {best[2]} reps in {best[3]:.1f}s")
    b1 = max(sweep.items(), key=lambda kv: kv[1]["b1_decisions_per_s"])
    out["gate_b1_decisions_per_s"] = b1[1]["b1_decisions_per_s"]
    out["gate_b1_ms"] = b1[1]["b1_ms"]
    out["gate_b1_threads"] = int(b1[0])
    out["gate_thread_sweep"] = sweep
    # ---- the C restatement (oracle/prag_oracle.c, float64 accumulators, OpenMP) ---------------------
    xs_np, q_np, x_np = x_c1, q_c1, x.numpy()
    r, dt = _timed(lambda: oracle_c.flat_search(xs_np, q_np, k1, 0), slice_s)
    c = {"threads": oracle_c.num_threads(), "flat_scores_per_s": 128 * N1 * r / dt}
    r, dt = _timed(lambda: oracle_c.flat_search(xs_np, q_np[:1], k1, 0), slice_s, 2000)
    c["flat_b1_scores_per_s"] = N1 * r / dt

    def c_gate(xb):
        lg = np.stack([oracle_c.prober_forward(states[l], xb[l]) for l in range(N_LAYERS)])
        return oracle_c.gate(lg, 0, 0.0)
    r, dt = _timed(lambda: c_gate(x_np), slice_s)
    c["gate_decisions_per_s"] = 128 * r / dt
    r, dt = _timed(lambda: c_gate(x_np[:, :1]), slice_s, 2000)
    c["gate_b1_decisions_per_s"] = r / dt
    out["c_restatement"] = c
    # ---- bounded sample of the metric's workload -------------------------------------------------
    Ns, Bq, k = 200_000, args.queries, args.k
    gq = torch.Generator().manual_seed(7)
    xs = torch.randn((Ns, D_EMB), generator=gq)
    qs = torch.randn((Bq, D_EMB), generator=gq)
    if args.metric == "cos":
        xs = torch.nn.functional.normalize(xs, dim=1)
    xn = (xs * xs).sum(1)
    r, dt = _timed(lambda: torch_cpu.flat_search(xs, xn, qs, k, args.metric == "l2"), slice_s * 1.5, 200)
    out["metric_workload_sample"] = {
        "value": Bq * Ns * r / dt, "unit": "query*doc scores/s",
        "sample": f"torch-cpu flat {args.metric} top-{k}: {Bq} queries x {Ns} docs x {D_EMB} fp32, {r} reps in {dt:.1f}s"}
    return out


# ---------------------------------------------------------------------------------------------
def _timed_searches(torch, ix, q, k, min_s=0.6, max_reps=200):
    """ms per search (host-timed over back-to-back searches) + the profiled scan-kernel launches."""
    for _ in range(10):             # (waited for: the scan workgroup count of a <= 64-query two-level search settles here)
        ix.search(q, k)
        torch.cuda.synchronize()
    ix.profile(1024)
    t0 = time.perf_counter()
    reps = 0
    while True:
        for _ in range(4):
            ix.search(q, k)
        reps += 4
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > min_s or reps >= max_reps:
            break
    dt = time.perf_counter() - t0
    ms = ix.profile_read()
    ix.profile(0)
    return dt / reps * 1e3, ms, ix.last_exact_fallbacks()


def scan_model(ix):
    """(kernel name, algorithmic bytes per launch, launches per search, tiled?) of the most recent search on `ix`,
    read from the plan the library executed (prag_index_last_plan: the dispatch is decided in ONE place,
    plan_search in flat_index.hip; round 3 re-derived it here).  Algorithmic bytes = what one pass over the shard
    has to read: the rows in the form that is scanned (+4 B of ||x||^2 per row for L2; the 8-bit shadow carries
    12 B per row: scale, error bound and the additive part of the key)."""
    plan = ix.last_plan()
    return plan["family"], plan["bytes_per_launch"], plan["launches"], bool(plan["tiled"])


def plan_summary(plan):
    """The launch-determining fields of a plan on record, as one short string (compared across ranks at N > 1)."""
    keys = ("family", "QT", "kc", "grid", "launches", "shadow", "tiled", "int8_tiles", "gate_launch", "adaptive")
    return " ".join(f"{k_}={plan[k_]}" for k_ in keys if k_ in plan)


def stored_row_bytes(store, metric, n_local):
    """N*d*s of the rows AS STORED (+4 B of ||x||^2 per row for L2)."""
    return n_local * D_EMB * (2 if store == "f16" else 4) + (n_local * 4 if metric == "l2" else 0)


ROOFLINE_DEFINITION = (
    "frac = frac_bytes_moved = bytes the scan kernel has to read per launch (the rows in the form it scans: "
    "the 8-bit shadow + 12 B/row of scale, error bound and additive key term for scan8_kernel, the stored rows otherwise) / kernel "
    "time / 8 TB/s.  frac_stored_rows = SURVEY 8(d)'s N*d*s of the rows as stored / the same kernel time / 8 TB/s: "
    "above the bytes-moved fraction (and possibly above 1) exactly when the two-level search avoids reading the "
    "stored rows; for a direct scan the two are equal.")


def variant_record(torch, ix, q, k, store, metric, n_local, shadow=0):
    B = q.shape[0]
    ix.set_shadow(1 if shadow else 0)
    ms_search, kern_ms, fb = _timed_searches(torch, ix, q, k)
    kernel, alg_bytes, passes, tiled = scan_model(ix)
    rec = {"store": store, "metric": metric, "k": k, "queries": B, "rows": n_local, "shadow": bool(shadow),
           "ms_per_search": ms_search, "scores_per_s": B * n_local / (ms_search * 1e-3),
           "exact_fallbacks_last_search": fb}
    if tiled:   # MFMA-bound: price the whole search (all segments, compactions, rerank) against the matrix peak
        tf = 2.0 * B * n_local * D_EMB / (ms_search * 1e-3) / 1e12
        i8 = bool(shadow) and ix.last_tiled8() >= 0
        rec.update({"kernel": kernel + " (whole search)", "bound": "mfma", "achieved": tf,
                    "peak": MFMA_I8_PEAK_TOP if i8 else MFMA_F16_PEAK_TF, "unit": "Top/s" if i8 else "TFLOP/s",
                    "frac": tf / (MFMA_I8_PEAK_TOP if i8 else MFMA_F16_PEAK_TF),
                    "frac_of_f16_peak": tf / MFMA_F16_PEAK_TF,
                    "largest_segment_ms": float(np.mean(kern_ms)) if kern_ms else None})
        if i8:
            rec["int8_tier_failed_last_search"] = ix.last_tiled8()
    else:
        kms = float(np.mean(kern_ms)) if kern_ms else float("nan")
        gbs = alg_bytes / (kms * 1e-3) / 1e9
        stored = stored_row_bytes(store, metric, n_local)
        rec.update({"kernel": kernel,
                    "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "frac_bytes_moved": gbs / HBM_PEAK_GBS,
                    "frac_stored_rows": stored / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "avg_launch_ms": kms, "launches_per_search": passes,
                    "algorithmic_bytes_per_launch": alg_bytes, "stored_row_bytes": stored,
                    "whole_search_frac": alg_bytes * passes / (ms_search * 1e-3) / 1e9 / HBM_PEAK_GBS})
    return rec


def _run_group(cmd, cwd=None, env=None, timeout=None):
    """Run `cmd` as a child in a session of its own and wait; on timeout kill the WHOLE process group (rocprofv3
    starts the Python program as a grandchild: killing the profiler alone left it holding the GPU and ~48 GB).
    Returns None on exit (any code) or a short error string."""
    try:
        proc = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                start_new_session=True)
    except Exception as e:
        return type(e).__name__
    try:
        proc.wait(timeout=timeout)
        return None
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return "timeout"


GATE_KERNELS = ("prober16_kernel", "prober_fused_kernel")


def _pmc_passes(counters, kernel_substr, prof_args):
    """One child `rocprofv3 --pmc <counter>` pass per counter (never combined with trace domains) over
    tools/prof_kernels.py; returns ({counter: mean value over the full-grid launches of the kernel} or None, note)."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    # a profiler's environment is inherited by children: never start a counter pass from inside a profiled run
    if any(k.startswith(("ROCP", "ROCPROF", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is being profiled: no nested rocprofv3 pass"
    got = {}
    for counter in counters:
        out_dir = tempfile.mkdtemp(prefix="prag_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--",
               sys.executable, os.path.join(ROOT, "tools", "prof_kernels.py")] + list(prof_args)
        err = _run_group(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, timeout=300)
        if err is not None:
            shutil.rmtree(out_dir, ignore_errors=True)
            return None, f"rocprofv3 --pmc {counter} failed: {err}"
        rows = []
        for f in glob.glob(os.path.join(out_dir, "**", "*_counter_collection.csv"), recursive=True):
            subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
            rows += [r for r in csv.DictReader(open(f))
                     if any(s_ in r["Kernel_Name"] for s_ in subs) and r["Counter_Name"] == counter]
        shutil.rmtree(out_dir, ignore_errors=True)
        if not rows:
            return None, f"no {kernel_substr} launch in the --pmc {counter} pass"
        full = max(int(r["Grid_Size"]) for r in rows)          # (the list scan's pre-pass runs on a few workgroups)
        vals = [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == full]
        got[counter] = sum(vals) / len(vals)
    return got, "measured by this run: child `rocprofv3 --pmc` passes over tools/prof_kernels.py, one counter per pass"


def measure_traffic(n_local, store, metric, queries, shadow, kernel, k=10):
    """HBM bytes per launch of `kernel` from two separate child `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE),
    corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE in KiB reports half of a wide streaming
    read; WRITE_SIZE in KiB is exact).  Returns (bytes or None, note)."""
    got, note = _pmc_passes(("FETCH_SIZE", "WRITE_SIZE"), kernel.split("_kernel")[0],
                            ["--skip-gate", "--docs", str(n_local), "--queries", str(queries), "--store", store,
                             "--metric", metric, "--shadow", str(1 if shadow else 0), "--iters", "3", "--k", str(k)])
    if got is None:
        return None, note
    return 2.0 * got["FETCH_SIZE"] * 1024 + got["WRITE_SIZE"] * 1024, \
        "measured by this run: child `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over tools/prof_kernels.py " \
        "(reads doubled per the gfx950 correction of MI355X_MICROARCH.md)"


def measure_gate_f32_pmc(gate_batch):
    """The reference-precision gate (fp32 states, hi+lo fp16 weight terms): HBM bytes and matrix-pipe utilisation of
    prober_fused_kernel from four child `rocprofv3 --pmc` passes.  Returns (dict or None, note)."""
    got, note = _pmc_passes(("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"), "prober_fused_kernel",
                            ["--skip-scan", "--gate-batch", str(gate_batch), "--iters", "5", "--weights", "f32",
                             "--x-dtype", "f32"])
    if got is None:
        return None, note
    return {"hbm_bytes_per_launch": 2.0 * got["FETCH_SIZE"] * 1024 + got["WRITE_SIZE"] * 1024,
            "matrix_pipe_busy_frac_of_cu_busy": (got["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * got["SQ_BUSY_CU_CYCLES"])
                                                 if got["SQ_BUSY_CU_CYCLES"] else None)}, note


def measure_gate_mfma(gate_batch):
    """Matrix-pipe utilisation of the fused prober kernel (prober16_kernel: 16x16x32 MFMA tiles, the default;
    prober_fused_kernel: the 32x32x16 form behind PRAG_PROBER_SHAPE=32): SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x
    SQ_BUSY_CU_CYCLES), each counter in its own child pass.  Returns (fraction or None, note)."""
    got, note = _pmc_passes(("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"), GATE_KERNELS,
                            ["--skip-scan", "--gate-batch", str(gate_batch), "--iters", "5"])
    if got is None or not got["SQ_BUSY_CU_CYCLES"]:
        return None, note
    return got["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * got["SQ_BUSY_CU_CYCLES"]), note


# ---------------------------------------------------------------------------------------------
# The line the driver reads.  Round 5's single line grew to 20.7 KB and the driver could not parse it: the LAST stdout
# line is now a compact record (< 6 KB, asserted by tests/test_bench_launch_cpu.py); the full record - variants,
# thread sweeps, notes - goes out on an EARLIER line prefixed `BENCH_DETAIL ` and into gpurun_out/bench_detail.json.
# ---------------------------------------------------------------------------------------------
COMPACT_LIMIT = 6000


def _sig(o, sig=5):
    """Floats to `sig` significant digits (the compact line is a summary; the detail record keeps every digit)."""
    if isinstance(o, bool) or o is None:
        return o
    if isinstance(o, float):
        if o != o or abs(o) == float("inf"):
            return None
        if o == 0.0:
            return 0.0
        if abs(o) >= 1e15:
            return o
        r = float(f"{o:.{sig}g}")
        return int(r) if r.is_integer() and abs(r) >= 10 ** sig else r
    if isinstance(o, dict):
        return {k: _sig(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_sig(v, sig) for v in o]
    return o


def _pick(d, *keys):
    d = d or {}
    return {k: d.get(k) for k in keys if d.get(k) is not None}


def compact_record(full):
    """The < 6 KB record printed as the last stdout line: the contract's keys, `config` as scalars only, `roofline`
    for the dominant kernel (+ the gate's kernel as `roofline.gate`), `cpu_baseline` with a short sample string."""
    cfg = dict(full.get("config") or {})
    emb = cfg.pop("embedding_like", None) or {}
    for name in ("cos_4M", "l2_4M", "cos_21M"):                     # three numbers, not three dicts
        e = emb.get(name) or {}
        if e:
            cfg[f"emb_{name}_two_level_ms"] = (e.get("two_level") or {}).get("ms_per_search")
            cfg[f"emb_{name}_ids_identical"] = e.get("ids_identical")
    cfg.pop("shard_pass_ms_by_mode", None)
    cfg.pop("gate_overlap", None)
    cfg = {k: v for k, v in cfg.items() if not isinstance(v, (dict, list))}
    roof = full.get("roofline") or {}
    r = _pick(roof, "bound", "kernel", "achieved", "peak", "unit", "frac", "frac_stored_rows", "traffic",
              "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "avg_launch_ms", "launches_per_pass",
              "launches_timed", "shard_scan8_frac", "shard_scan8_ms")
    r.setdefault("traffic", None)
    sa = roof.get("scan_alone") or {}
    if sa:
        r["scan_alone_ms"], r["scan_alone_frac"] = sa.get("avg_launch_ms"), sa.get("frac")
    g = full.get("roofline_gate") or roof.get("gate") or {}
    if g:
        r["gate"] = _pick(g, "bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "hbm_GBs")
        r["gate"]["matrix_pipe_busy"] = g.get("matrix_pipe_busy_frac_of_cu_busy")
    cb = full.get("cpu_baseline")
    if cb:
        c = _pick(cb, "cores", "kind", "unit", "value", "gate_decisions_per_s", "gate_threads", "gate_b1_decisions_per_s",
                  "flat_b1_scores_per_s")
        c["sample"] = (cb.get("sample_short") or cb.get("sample") or "")[:100]
        mw = cb.get("metric_workload_sample") or {}
        if mw:
            c["metric_workload_scores_per_s"] = mw.get("value")
        cb = c
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "backend", "steps", "warmup",
                                    "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["dtype"] = (full.get("dtype_short") or str(full.get("dtype") or ""))[:24]
    out["config"] = cfg
    out["roofline"] = r
    out["cpu_baseline"] = cb
    for k in ("probe_decisions_per_s", "scores_per_s_per_gpu", "planted_top1_recall", "result_lists_sorted",
              "exact_fallbacks_last_search", "recall_at_k_vs_oracle", "topk_ids_bit_exact_vs_oracle", "exchange",
              "plan", "plans_identical_across_ranks", "detail"):
        if full.get(k) is not None:
            out[k] = full[k]
    pr = full.get("per_rank")
    if pr:                                   # N > 1: one short row per rank
        out["per_rank"] = [[p.get("rows"), p.get("scan_ms"), p.get("gate_ms"), p.get("allgather_us")] if p else None
                           for p in pr]
        out["per_rank_columns"] = "rows, scan_ms, gate_ms, allgather_us"
    cp = full.get("c_exchange_probe")
    if cp:
        out["c_exchange_probe"] = _pick(cp, "ok", "ms_per_pass", "allgather_us", "error")
    out = _sig(_finite(out))
    for k in ("value", "ms_per_step"):       # the driver checks these against its own clock: every digit
        out[k] = _finite(full.get(k))
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:            # never let a long string through: drop optional blocks until it fits
        for k in ("per_rank", "c_exchange_probe", "plan", "per_rank_columns"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= COMPACT_LIMIT:
                break
    return out, line


def emit(full, stream=None):
    """BENCH_DETAIL line + side file first, the compact line LAST (the only line that starts with `{`)."""
    stream = stream or sys.stdout
    full = _finite(full)
    detail_path = None
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        detail_path = os.path.join("gpurun_out", "bench_detail.json")
        with open(os.path.join(ROOT, detail_path), "w") as f:
            json.dump(full, f)
    except OSError:
        detail_path = None
    full["detail"] = "BENCH_DETAIL line above" + (f"; {detail_path}" if detail_path else "")
    _, line = compact_record(full)
    stream.write("BENCH_DETAIL " + json.dumps(full) + "\n")
    stream.write(line + "\n")
    stream.flush()


def quiet_stdout():
    """Library chatter (gloo / RCCL banners, rocm warnings) must not land on stdout next to the record: fd 1 is
    pointed at stderr for the rest of the process and the record is written to a private copy of the real stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def launch_command(n_gpus, argv, port=None):
    """The one-rank-per-GPU launch of this file: the command the driver documents, on a free local port."""
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher: start `torch.distributed.run` as a fresh CHILD
    process - before torch is imported or any HIP call is made here, and never by exec (a process that has
    touched the GPU must not be replaced) - let its output through and return its exit code.
    Returns None when this process is itself a rank (WORLD_SIZE set) or N == 1."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    assert "torch" not in sys.modules, "the launcher must not have imported torch"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = launch_command(args.gpus, argv)
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    timeout = getattr(args, "launch_timeout", None)
    if not timeout or timeout <= 0:
        return subprocess.run(cmd, env=env).returncode
    # a hung rendezvous or collective must not hang the caller: the child gets a session of its own, and on timeout
    # its whole process group (the launcher and every rank) is killed - never this process replaced or re-executed
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {args.gpus}-rank child did not finish within {timeout:.0f} s: killing its process group",
              file=sys.stderr, flush=True)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=15)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


def clustered_variant(torch, pra, d_emb, k, n_rows=4_194_304, n_centres=4096, sigma=0.0175, B=64, interleaved=False):
    g = torch.Generator(device="cuda").manual_seed(11)
    centres = torch.nn.functional.normalize(torch.randn((n_centres, d_emb), generator=g, device="cuda"), dim=1)
    ix = pra.HipFlatIndex(d_emb, "cos", "f16", capacity=n_rows)
    for lo in range(0, n_rows, 1 << 20):
        m = min(1 << 20, n_rows - lo)
        # rows of a centre are CONTIGUOUS (a corpus in article order: consecutive passages resemble each other),
        # so a centre's ~1000 rows fall into a handful of scan workgroups - the layout that fills a region
        # interleaved: every row draws its centre at random - a clustered corpus in random order, a centre's rows spread
        # over every scan workgroup.  (`i mod n_centres` was the first layout tried: a period of 4096
        # rows = 128 scan tiles resonated with a 256-workgroup grid - all 1024 rows of a centre in TWO workgroups' candidate
        # regions, 62 of 64 queries flagged, 2.2 ms per search - profiles/r06i_clustered_interleaved.txt; the library's own
        # grids are prime counts since, 0.69 ms on that layout - profiles/r06r_prime_grid_ab.txt.)
        idx = torch.randint(0, n_centres, (m,), generator=g, device="cuda") if interleaved else \
            (torch.arange(lo, lo + m, device="cuda") * n_centres) // n_rows
        ix.add(centres[idx] + sigma * torch.randn((m, d_emb), generator=g, device="cuda"))   # ||noise|| ~ 0.48
    q = centres[torch.randint(0, n_centres, (B,), generator=g, device="cuda")] + \
        0.5 * sigma * torch.randn((B, d_emb), generator=g, device="cuda")
    rec = {"rows": n_rows, "centres": n_centres, "queries": B, "k": k,
           "what": "rows = unit centre + N(0, sigma^2 I), sigma = %.4f: 1024 %s rows per centre at cosine ~0.9 to it"
                   % (sigma, "INTERLEAVED (every row draws its centre at random)" if interleaved else "CONTIGUOUS")}
    res = {}
    for shadow in (0, 2):
        ix.set_shadow(shadow)
        ix.prepare()
        ms, kern, fb = _timed_searches(torch, ix, q, k, min_s=0.3, max_reps=100)
        res[shadow] = ix.search(q, k)
        rec["two_level" if shadow else "rows_scanned_directly"] = {
            "ms_per_search": ms, "scan_kernel_ms": float(np.mean(kern)) if kern else None, "exact_fallbacks_last_search": fb}
    rec["ids_identical"] = bool(torch.equal(res[0][1], res[2][1]))
    ix.close()
    return rec


def embedding_variant(torch, pra, d_emb, k, n_rows, metric="cos", store="f16", B=64, seed=5, min_s=0.3, **structure_kw):
    """A corpus with the geometry of real sentence embeddings (probing_rag_amd/synth.py
    embedding_like_rows: common mean at 0.8 of the row norm, power-law spectrum, six outlier coordinates at 10-30 x the
    median |x_j|, un-normalised) - the reference indexes un-normalised contriever output under L2
    (make_indexer.py:447-456).  Queries come from the same distribution.  Two-level search next to the direct scan of
    the stored rows: time, exact fallbacks, ids identical."""
    from probing_rag_amd.synth import embedding_like_rows, embedding_structure
    st = embedding_structure(seed, d_emb, **structure_kw)
    ix = pra.HipFlatIndex(d_emb, metric, store, capacity=n_rows)
    step = 1 << 19
    for lo in range(0, n_rows, step):
        ix.add(embedding_like_rows(seed, lo, min(step, n_rows - lo), d_emb, structure=st))
    q = embedding_like_rows(seed + 1000, 0, B, d_emb, structure=st)
    rec = {"rows": n_rows, "queries": B, "k": k, "metric": metric, "store": store}
    res = {}
    for shadow, name in ((0, "rows_scanned_directly"), (2, "two_level")):
        ix.set_shadow(shadow)
        ix.prepare()
        ms, kern, fb = _timed_searches(torch, ix, q, k, min_s=min_s, max_reps=100)
        res[shadow] = ix.search(q, k)
        rec[name] = {"ms_per_search": ms, "scan_kernel_ms": float(np.mean(kern)) if kern else None,
                     "exact_fallbacks_last_search": fb, "kernel": ix.last_plan().get("family")}
        if shadow and hasattr(ix, "last_survivors"):
            rec[name]["survivors"] = ix.last_survivors()
    rec["ids_identical"] = bool(torch.equal(res[0][1], res[2][1]))
    rec["two_level_over_direct"] = rec["two_level"]["ms_per_search"] / rec["rows_scanned_directly"]["ms_per_search"]
    ix.close()
    return rec


SHARD_ROWS, SHARD_QUERIES, SHARD_GATE_ROWS = 2_625_000, 64, 512     # one rank's share of the headline at 8 GPUs


def shard_variant(torch, pra, ens, k, metric, ms_per_pass_full, passes=300, adaptive=False):
    """The 8-GPU shard shape on this one GPU: what ONE rank of the 8-rank job runs per pass - the gate over 512 of
    the 4096 pooled states, then the top-k of the 64 replicated queries over its 2 625 000 rows (the all-gather of
    8 x 64 x 10 x 12 bytes and the merge are not in it).  Timed four ways (same inputs, same outputs checked):
      eager        gate then search on one stream, one host call each (rounds 1-4)
      graph        the same pass captured into ONE HIP graph and replayed (every piece is capturable: no host
                   synchronisation, no allocation after the first pass of a shape)
      tail_2s      the gate of the NEXT batch of states (it depends on nothing in the search) on a second stream that waits
                   for the search's CORPUS SCAN only (prag_index_stream_wait_scan): its 96 workgroups run beside the
                   search's tail - the 64-workgroup bound kernel, the rerank, the fallback probes.  (Beside the scan
                   itself it cannot run: scan8 holds all of a CU's LDS.  Round 5 measured that form too - the gate issued
                   at the head of the pass on a second stream: 0.472 eager / 0.519 captured against 0.465 on one stream,
                   profiles/r05a_bench.json.)
      graph_tail_2s  the same, captured (fork / join inside the graph)
      fused        prag_search_and_gate: one C call; the gate's prober workgroups ride in the launch of the search's bound
                   kernel (bound_gate_kernel) - no second stream, no cross-stream dependency
      graph_fused  the same, captured
    pass_ms = the fastest of them (named in pass_mode).  predicted_strong_scaling_eff = ms_per_pass(21 M rows, this
    run) / 8 / pass_ms: the efficiency an 8-GPU run can reach at best (fixed costs do not shrink with the shard)."""
    from probing_rag_amd.synth import synth_rows
    d_emb, d_model, L = D_EMB, D_MODEL, N_LAYERS
    ix = pra.HipFlatIndex(d_emb, metric, "f16", capacity=SHARD_ROWS)
    ix.add_synthetic(42, 0, SHARD_ROWS)
    ix.set_shadow(1)
    ix.set_adaptive(adaptive)
    ix.prepare()
    q = torch.from_numpy(synth_rows(7, 0, SHARD_QUERIES, d_emb)).cuda()
    g = torch.Generator(device="cuda").manual_seed(4321)
    x = torch.randn((L, SHARD_GATE_ROWS, d_model), generator=g, device="cuda", dtype=torch.float32).half()
    gate_out = (torch.empty((L, SHARD_GATE_ROWS, 2), dtype=torch.float32, device="cuda"),
                torch.empty((SHARD_GATE_ROWS, 2), dtype=torch.float32, device="cuda"),
                torch.empty((SHARD_GATE_ROWS,), dtype=torch.int32, device="cuda"))
    out = (torch.empty((SHARD_QUERIES, k), dtype=torch.float32, device="cuda"),
           torch.empty((SHARD_QUERIES, k), dtype=torch.int64, device="cuda"))
    ix.reserve(SHARD_QUERIES, k)
    ens.reserve(SHARD_GATE_ROWS)

    def timed(fn, n):
        for _ in range(12):         # (waited for one by one: the index settles its scan workgroup count on these)
            fn()
            torch.cuda.synchronize()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def one_pass():
        ens.gate(x, 0, 0.0, out=gate_out)
        ix.search(q, k, out=out)

    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()

    def one_pass_tail():
        ix.search(q, k, out=out)
        ix.stream_wait_scan(side)                    # the side stream waits for scan8, not for the whole search
        with torch.cuda.stream(side):
            ens.gate(x, 0, 0.0, out=gate_out)
        torch.cuda.current_stream().wait_stream(side)

    def one_pass_fused():                            # one C call: the gate's workgroups in the bound kernel's launch
        pra.search_and_gate(ix, q, k, ens, x, 0, 0.0, out=out, gate_out=gate_out)

    ix.stream_wait_scan(side)                        # (first call: switches the event recording on)

    modes = {}
    modes["eager"] = timed(one_pass, passes)                       # no event rings inside the timed loop
    I_eager, dec_eager = out[1].clone(), gate_out[2].clone()
    search_ms = timed(lambda: ix.search(q, k, out=out), passes)
    gate_ms = timed(lambda: ens.gate(x, 0, 0.0, out=gate_out), passes)
    modes["tail_2s"] = timed(one_pass_tail, passes)
    same = {"tail_2s": bool(torch.equal(out[1], I_eager) and torch.equal(gate_out[2], dec_eager))}
    out[1].zero_()
    gate_out[2].zero_()
    modes["fused"] = timed(one_pass_fused, passes)
    same["fused"] = bool(torch.equal(out[1], I_eager) and torch.equal(gate_out[2], dec_eager))
    graph_err = None
    for name, fn in (("graph", one_pass), ("graph_tail_2s", one_pass_tail), ("graph_fused", one_pass_fused)):
        try:
            cs = torch.cuda.Stream()
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                fn()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=cs):
                    fn()
            torch.cuda.synchronize()
            out[1].zero_()
            gate_out[2].zero_()
            modes[name] = timed(gr.replay, passes)
            same[name] = bool(torch.equal(out[1], I_eager) and torch.equal(gate_out[2], dec_eager))
            del gr
        except Exception as e:      # noqa: BLE001 - a capture problem must not take the bench down
            graph_err = f"{name}: {type(e).__name__}: {e}"
            torch.cuda.synchronize()
    valid = {m: t for m, t in modes.items() if same.get(m, True)}
    pass_mode = min(valid, key=valid.get)
    pass_ms = valid[pass_mode]
    ix.profile(256)
    ens.profile(256)
    for _ in range(200):
        one_pass()
    torch.cuda.synchronize()
    scan_ms = ix.profile_read()
    gk_ms = ens.profile_read()
    ix.profile(0)
    ens.profile(0)
    fb = ix.last_exact_fallbacks()
    I_two_level = out[1].clone()
    ix.set_shadow(0)
    _, I_direct = ix.search(q, k)
    scan_k = float(np.mean(scan_ms)) if scan_ms else float("nan")
    gate_k = float(np.mean(gk_ms)) if gk_ms else float("nan")
    alg = SHARD_ROWS * (d_emb + 12)
    rec = {"what": "one rank's pass of the 8-GPU job on this GPU: gate over 512 x 6 x 2048 fp16 states, then %s top-%d of "
                   "64 queries over 2 625 000 x 768 fp16 rows (two-level search); host-timed back to back, "
                   "no event records in the timed loops" % (metric, k),
           "rows": SHARD_ROWS, "queries": SHARD_QUERIES, "gate_rows": SHARD_GATE_ROWS, "k": k,
           "pass_ms": pass_ms, "pass_mode": pass_mode, "pass_ms_by_mode": modes, "outputs_identical_by_mode": same,
           "graph_error": graph_err,
           "search_alone_ms": search_ms, "gate_alone_ms": gate_ms,
           "scan8_kernel_ms_in_pass": scan_k, "gate_kernel_ms_in_pass": gate_k,
           "rest_of_pass_ms (prep, gather + merge, exact probe, launch gaps)": modes["eager"] - scan_k - gate_k,
           "scan8_frac_of_8TBs": alg / (scan_k * 1e-3) / 1e9 / HBM_PEAK_GBS if scan_k == scan_k else None,
           "exact_fallbacks_last_search": fb,
           "ids_identical_to_direct_scan_of_the_rows": bool(torch.equal(I_two_level, I_direct) and torch.equal(I_eager, I_direct)),
           "ms_per_pass_21M_this_run": ms_per_pass_full,
           "predicted_strong_scaling_eff": ms_per_pass_full / 8.0 / pass_ms,
           "predicted_strong_scaling_eff_eager": ms_per_pass_full / 8.0 / modes["eager"]}
    ix.close()
    return rec


def c_exchange_probe(torch, dist, index, local, ens, x, gate_out, q, args, I_want, fence, rank, passes=100, limit_s=120.0):
    """First contact of the C-level exchange with more than one GPU (collective; every rank calls it): the library's
    own RCCL communicator, then `passes` passes of gate + prag_index_search_sharded timed like the headline.  A
    watchdog thread ends the WHOLE rank (os._exit) when the probe does not finish within `limit_s` - a collective that
    a peer never enters cannot be interrupted from Python - after rank 0 has printed what it knows, so the run still
    leaves a record.  The headline numbers are final before this is called."""
    import threading
    rec = {"ok": False}
    state = {"line": None}

    def bail():
        if rank == 0:
            print(json.dumps({"c_exchange_probe_watchdog": "the probe did not finish within %.0f s; rank ended" % limit_s,
                              "stage": rec.get("stage")}), file=sys.stderr, flush=True)
        os._exit(3)

    dog = threading.Timer(limit_s, bail)
    dog.daemon = True
    dog.start()
    try:
        rec["stage"] = "enable"
        ok, why = index.enable_c_exchange()
        if not ok:
            rec["error"] = why
            return rec
        rec["stage"] = "warm"
        for _ in range(3):
            ens.gate(x, 0, 0.0, out=gate_out)
            D, I = index.search(q, args.k)
        fence()
        rec["ids_identical_to_torch_distributed_exchange"] = bool(torch.equal(I, I_want))
        rec["stage"] = "timed"
        local.profile(min(4096, passes + 8))
        fence()
        t0 = time.perf_counter()
        for _ in range(passes):
            ens.gate(x, 0, 0.0, out=gate_out)
            index.search(q, args.k)
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        xms = local.profile_read_exchange()
        local.profile_read()
        local.profile(0)
        rec.update({"ok": True, "ms_per_pass": float(t.item()) / passes * 1e3, "passes": passes,
                    "allgather_us": float(np.mean(xms)) * 1e3 if xms else None})
        rec["stage"] = "disable"
        torch.cuda.synchronize()
        index.disable_c_exchange()
        rec.pop("stage", None)
        return rec
    except Exception as e:      # noqa: BLE001 - a probe must never take the headline down
        rec["error"] = f"{type(e).__name__}: {e}"
        return rec
    finally:
        dog.cancel()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    rc = self_launch(args, argv)
    if rc is not None:
        raise SystemExit(rc)
    real_stdout = quiet_stdout()        # from here on only emit() reaches the real stdout
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: using the launcher's world size",
              file=sys.stderr, flush=True)
        args.gpus = world
    n_dev = torch.cuda.device_count()            # does not initialise the GPU
    # RCCL ("nccl") is the product path.  PRAG_BENCH_BACKEND=gloo exists only to exercise the
    # multi-rank control flow on a box with fewer GPUs than ranks (collectives staged via host,
    # several ranks share a device).
    backend = os.environ.get("PRAG_BENCH_BACKEND", "nccl") if world > 1 else "none"
    if world > 1 and backend == "nccl" and n_dev < world:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, {n_dev} visible "
                         f"(PRAG_BENCH_BACKEND=gloo runs the rank logic on fewer)")
    dev_index = local_rank % max(1, n_dev)
    torch.cuda.set_device(dev_index)
    rank_devices = [dev_index]
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
        backend = dist.get_backend()
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, dev_index)

    import probing_rag_amd as pra
    from probing_rag_amd.synth import random_prober_state, synth_rows

    if args.e2e:
        import bench_e2e
        out = bench_e2e.run(args, pra, torch, dist, world, rank, dev_index)
        if rank == 0:
            out.update({"rccl_ranks": (dist.get_world_size() if world > 1 else 1), "backend": backend,
                        "rank_devices": rank_devices})
            real_stdout.write(json.dumps(_finite(out)) + "\n")
            real_stdout.flush()
        if world > 1:
            dist.destroy_process_group()
        return

    d_model, d_emb, L = D_MODEL, D_EMB, N_LAYERS
    # ---- corpus shard (generated on device by the counter-based generator of add_synthetic)
    lo, hi = pra.partition_rows(args.docs, world, rank)
    n_local = hi - lo
    index = pra.ShardedFlatIndex(d_emb, args.metric, args.store, capacity=n_local)
    index.add_synthetic_local(42, lo, n_local)
    index.sync()
    local = index.engine.index
    # ---- gate: 6 probers (random init of the reference architecture), this rank's slice of the states
    states = [random_prober_state(100 + l, d_model) for l in range(L)]
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
    q_np = synth_rows(7, 0, args.queries, d_emb)
    n_plant = min(8, args.queries)
    planted = [(i * 2_654_435 + 17) % args.docs for i in range(n_plant)]
    for i, r in enumerate(planted):
        q_np[i] = synth_rows(42, r, 1, d_emb)[0] + 0.05 * q_np[i]
    q = torch.from_numpy(q_np).cuda()

    main_stream = torch.cuda.current_stream()
    side_stream = torch.cuda.Stream()
    n_cu = torch.cuda.get_device_properties(dev_index).multi_processor_count
    if args.overlap_gate == 1:
        local.set_scan_workgroups(n_cu - 16)
    elif args.scan_workgroups:
        local.set_scan_workgroups(args.scan_workgroups)
    local.set_shadow(1 if args.shadow else 0)
    local.set_adaptive(bool(args.adaptive))

    if args.overlap_gate >= 2:
        local.stream_wait_scan(side_stream)       # (first call: switches the event recording on)

    def one_pass():
        if not args.overlap_gate:
            ens.gate(x, 0, 0.0, out=gate_out)
            return index.search(q, args.k)
        if args.overlap_gate == 3:
            # one C call per rank: the gate's prober workgroups in the launch of the local search's bound kernel
            # (prag_search_and_gate), then the exchange
            out, _ = index.search_and_gate(q, args.k, ens, x, 0, 0.0, gate_out=gate_out)
            return out
        if args.overlap_gate >= 2:
            # the search first; the gate (independent work: the decisions of the NEXT batch of generations) starts on
            # the side stream as soon as the search's corpus scan is done and runs beside its low-occupancy tail
            out = index.search(q, args.k)
            local.stream_wait_scan(side_stream)
            with torch.cuda.stream(side_stream):
                ens.gate(x, 0, 0.0, out=gate_out)
            main_stream.wait_stream(side_stream)      # the pass ends when both are done
            return out
        # the search goes first so the scan's persistent workgroups settle on their CUs; the gate
        # (independent work: the decisions of the NEXT batch) fills the CUs left free
        start = torch.cuda.Event()
        start.record(main_stream)                 # everything before this pass
        out = index.search(q, args.k)
        side_stream.wait_event(start)             # NOT the search just enqueued
        with torch.cuda.stream(side_stream):
            ens.gate(x, 0, 0.0, out=gate_out)
        main_stream.wait_stream(side_stream)      # the pass ends when both are done
        return out

    def step():
        for _ in range(args.inner):
            out = one_pass()
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # --adaptive 1: on its first searches an index measures whether its scan runs better on 7/8 of the CUs or on all of
    # them (eight samples, each read when a later search finds it finished): waited-for passes let that settle before
    # the warm-up.  The deterministic plan (default) has nothing to settle
    for _ in range(12 if args.adaptive else 2):
        one_pass()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    n_pass = args.steps * args.inner
    slots = min(4096, n_pass * (1 + (args.queries - 1) // 64) + 8)
    local.profile(slots)
    ens.profile(min(4096, n_pass + 8))
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        D, I = step()
    fence()
    dt = time.perf_counter() - t0
    scan_ms = local.profile_read()
    gate_ms = ens.profile_read()
    local.profile(0)
    ens.profile(0)
    fallbacks = local.last_exact_fallbacks()
    plan_timed = plan_summary(local.last_plan())        # the plan of the timed passes (later searches overwrite the record)
    xch_ms = local.profile_read_exchange() if index.exchange == "prag_rccl" else []
    dt_rank = dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    per_rank = None
    if world > 1:
        # the exchange step by itself, outside the timed region, for the torch.distributed path (its collective runs on
        # torch's own stream; the current stream waits for it, so events on the current stream bracket it)
        allgather_us = float(np.mean(xch_ms)) * 1e3 if xch_ms else None
        if index.exchange == "torch.distributed":
            buf, _, _, gathered = index.engine.search_packed(q, args.k, index.id_offset, world)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for _ in range(5):
                dist.all_gather_into_tensor(gathered, buf)
            fence()
            for a_, b_ in ev:
                a_.record()
                dist.all_gather_into_tensor(gathered, buf)
                b_.record()
            fence()
            allgather_us = float(np.median([a_.elapsed_time(b_) for a_, b_ in ev])) * 1e3
        mine = {"rank": rank, "device": dev_index, "rows": n_local, "gate_rows": Bg, "plan": plan_timed,
                "scan_ms": float(np.mean(scan_ms)) if scan_ms else None,
                "gate_ms": float(np.mean(gate_ms)) if gate_ms else None,
                "allgather_us": allgather_us, "timed_region_s": dt_rank,
                "exact_fallbacks_last_search": fallbacks}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    c_probe = None
    if world > 1 and args.c_exchange_probe and backend == "nccl" and index.exchange == "torch.distributed":
        c_probe = c_exchange_probe(torch, dist, index, local, ens, x, gate_out, q, args, I, fence, rank)

    # ---- correctness riders (outside the timed region) -------------------------
    I_host = I.cpu().numpy()
    D_host = D.cpu().numpy()
    planted_ok = all(int(I_host[i, 0]) == planted[i] for i in range(n_plant))
    sorted_ok = bool((np.diff(D_host, axis=1) >= 0).all() if args.metric == "l2" else (np.diff(D_host, axis=1) <= 0).all())

    if rank != 0:
        if world > 1:
            index.close()
            dist.destroy_process_group()
        return

    # top-k recall against the oracle (exact float64 brute force, C) on a bounded sub-corpus searched
    # by the same kernels: first 200k rows, 16 of the queries
    from oracle import oracle_c
    n_sub = min(200_000, args.docs)
    sub = pra.HipFlatIndex(d_emb, args.metric, args.store, capacity=n_sub)
    sub.add_synthetic(42, 0, n_sub)
    sub.set_shadow(2 if args.shadow else 0)        # the rider checks the path the headline ran
    qs = q[: min(16, args.queries)]
    _, I_sub = sub.search(qs, args.k)
    metric_id = {"l2": 0, "ip": 1, "cos": 2}[args.metric]
    _, I_ref = oracle_c.flat_search(sub.reconstruct_n(0, n_sub), qs.cpu().numpy(), args.k, metric_id)
    I_sub = I_sub.cpu().numpy()
    recall = float(np.mean([len(set(a) & set(b)) / args.k for a, b in zip(I_sub.tolist(), I_ref.tolist())]))
    exact_order = bool(np.array_equal(I_sub, I_ref))
    sub.close()

    passes_total = args.steps * args.inner
    scores = args.queries * args.docs
    value = scores * passes_total / dt
    # which scan kernel served the pass
    scan_kernel, alg_bytes, launches, tiled = scan_model(local)     # the plan the timed searches executed
    # --overlap-gate 3: which launch carried the gate's prober workgroups (the scan's own - scan8_gate_kernel - when the
    # gate fits under the scan on the CUs the scan leaves free, else the bound kernel's)
    gate_launch = local.last_plan().get("gate_launch")
    scan_launch_name = "scan8_gate_kernel" if gate_launch == "scan8_gate_kernel" else scan_kernel
    scan_avg_ms = float(np.mean(scan_ms)) if scan_ms else float("nan")
    if per_rank:        # N > 1: the roofline is the SLOWEST rank's launch (every pass waits for it in the all-gather)
        slow = max((r for r in per_rank if r and r["scan_ms"]), key=lambda r: r["scan_ms"], default=None)
        if slow is not None:
            scan_avg_ms = slow["scan_ms"]
            alg_bytes = alg_bytes * slow["rows"] // max(1, n_local) if slow["rows"] != n_local else alg_bytes
    achieved = alg_bytes / (scan_avg_ms * 1e-3) / 1e9
    mm_tf = mm_flops = rows_last = None
    if tiled:
        # the profiled launch is the last (largest) corpus segment: its row count comes from the plan the library
        # executed (ONE place decides the segment schedule: plan_search / mm_segment_growth in flat_index.hip)
        i8_tiles = local.last_tiled8() >= 0
        rows_last = int(local.last_plan().get("last_seg_rows") or n_local)
        mm_flops = 2.0 * args.queries * rows_last * d_emb
        mm_tf = mm_flops / (scan_avg_ms * 1e-3) / 1e12
    # HBM traffic per launch comes from a separate `rocprofv3 --pmc` pass of this command (counters cannot be
    # read inside the run): the committed summary is quoted when it matches this kernel and shard size
    traffic, traffic_source = None, None
    for name in ("pmc_scan8.json", "pmc_scan_topk.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if traffic is None and os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("rows_per_launch") == n_local and rec.get("store") == args.store and rec.get("kernel") == scan_kernel:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/{name} (separate --pmc pass, not measured in this run)"
            except Exception:
                traffic = None
    if args.measure_traffic and world == 1 and not tiled:
        t_meas, note = measure_traffic(n_local, args.store, args.metric, args.queries, args.shadow, scan_kernel, args.k)
        if t_meas is not None:
            traffic, traffic_source = t_meas, note
        else:
            traffic_source = (traffic_source + "; " if traffic_source else "") + "not measured in this run: " + note
    gate_mfma_busy, gate_mfma_note = None, "not measured (--measure-traffic 0 or more than one rank)"
    if args.measure_traffic and world == 1 and not args.no_variants:     # (shard / diagnostic runs skip the two passes)
        gate_mfma_busy, gate_mfma_note = measure_gate_mfma(Bg)
    # the same kernel with the chip to itself (in the pass it runs beside the search's tail when --overlap-gate 2)
    ens.profile(64)
    for _ in range(50):
        ens.gate(x, 0, 0.0, out=gate_out)
    torch.cuda.synchronize()
    gate_alone = ens.profile_read()
    ens.profile(0)
    gate_alone_ms = float(np.mean(gate_alone)) if gate_alone else None
    # ... and the scan launch of a plain search (no gate in it), same index, same queries
    scan_alone_ms = None
    if world == 1 and not tiled:
        local.profile(64)
        for _ in range(40):
            local.search(q, args.k)
        torch.cuda.synchronize()
        sa = local.profile_read()
        local.profile(0)
        scan_alone_ms = float(np.sum(sa)) / 40 if sa else None
    # --overlap-gate 3: the prober's workgroups ride in bound_gate_kernel's launch, so the pass holds no prober launch
    # to put events around; the gate's own figures are then the stand-alone launch's (the fused launch's duration is
    # in the rocprofv3 summary of this command: bound_gate_kernel)
    gate_in_pass = bool(gate_ms)
    gate_avg_ms = float(np.mean(gate_ms)) if gate_ms else (gate_alone_ms if gate_alone_ms else float("nan"))
    gate_timing = ("HIP events around the prober's launches inside the timed region" if gate_in_pass else
                   "stand-alone launches after the timed region (in the pass the prober's workgroups ride in %s's launch)"
                   % (gate_launch or "bound_gate_kernel"))
    gate_flops = 2.0 * L * (d_model * 512 + 512 * 512 + 512 * 2) * Bg
    out = {
        "metric": "probe-decisions/sec + query*doc scores/sec/GPU (value = query*doc scores/sec, whole job)",
        "value": value, "unit": "query*doc scores/s",
        "n_gpus": world, "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "backend": backend,
        "rank_devices": rank_devices, "exchange": index.exchange, "exchange_note": index.exchange_note,
        "plan": plan_timed,
        "plans_identical_across_ranks": (len({r_["plan"] for r_ in per_rank if r_}) == 1) if per_rank else None,
        "dtype_short": "i8" if (scan_kernel == "scan8_kernel" or (tiled and i8_tiles)) else "f16",
        "per_rank": per_rank, "c_exchange_probe": c_probe,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("i8 shadow scan (MFMA i8, i32 accumulate) over f16 rows; f64 exact rerank of the filter's survivors"
                  if scan_kernel == "scan8_kernel" else
                  "i8 tiles over the 8-bit shadow (MFMA i8, i32 accumulate; 256 candidates per query) over f16 rows; f64 rerank + "
                  "exactness certificate" if (tiled and i8_tiles) else
                  "f16 (MFMA, f32 accumulate; f64 rerank + exactness certificate)"),
        "data": "synthetic",
        "config": {"workload": f"gate B={args.gate_batch} x 6 layers x d_model=2048 fp16 + flat {args.metric} "
                               f"top-{args.k} of {args.queries} queries over {args.docs} x 768 {args.store} docs",
                   "passes_per_step": args.inner, "ms_per_pass": ms_per_step / args.inner,
                   "timed_region_s": dt,
                   "docs_total": args.docs, "docs_per_gpu": n_local, "d_emb": d_emb, "queries": args.queries,
                   "k": args.k, "gate_batch": args.gate_batch, "gate_batch_per_gpu": Bg, "d_model": d_model,
                   "parallelism": f"corpus rows sharded x{world}; gate rows split x{world}",
                   "gate_overlap": {0: "none (one stream)", 1: "beside the scan", 2: "beside the search's tail (second stream waits for the scan only)", 3: "in a launch of the search (prag_search_and_gate): the scan's when the gate fits under it, else the bound kernel's"}[args.overlap_gate],
                   "two_level_shadow": scan_kernel == "scan8_kernel"},
        "probe_decisions_per_s": args.gate_batch / (gate_avg_ms * 1e-3) if gate_avg_ms == gate_avg_ms else None,
        "scores_per_s_per_gpu": value / world,
        "planted_top1_recall": 1.0 if planted_ok else 0.0, "result_lists_sorted": sorted_ok,
        "exact_fallbacks_last_search": fallbacks,
        "recall_at_k_vs_oracle": recall, "topk_ids_bit_exact_vs_oracle": exact_order,
        "recall_sample": f"{min(16, args.queries)} queries x {n_sub} docs, float64 C oracle",
        "roofline": ({"bound": "mfma", "kernel": "scan_mm_kernel<int8 tiles>" if i8_tiles else "scan_mm_kernel",
                      "achieved": mm_tf, "peak": MFMA_I8_PEAK_TOP if i8_tiles else MFMA_F16_PEAK_TF,
                      "unit": "Top/s" if i8_tiles else "TFLOP/s",
                      "frac": mm_tf / (MFMA_I8_PEAK_TOP if i8_tiles else MFMA_F16_PEAK_TF), "traffic": None,
                      "algorithmic_flops_per_launch": mm_flops, "rows_in_launch": rows_last,
                      "avg_launch_ms": scan_avg_ms, "launches_per_pass": launches} if tiled else
                     {"bound": "hbm", "kernel": scan_launch_name,
                      "launch_carries": ("the scan's workgroups (scan8_kernel's body) on 7/8 of the CUs and, behind them, the "
                                         "gate's prober workgroups on the rest: the duration is the whole launch's, the bytes "
                                         "are the scan's" if scan_launch_name == "scan8_gate_kernel" else "the scan alone"),
                      "achieved": achieved, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                      "frac_bytes_moved": achieved / HBM_PEAK_GBS,
                      "frac_stored_rows": stored_row_bytes(args.store, args.metric, n_local) / (scan_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "definition": ROOFLINE_DEFINITION,
                      "traffic": traffic, "traffic_source": traffic_source,
                      "algorithmic_bytes_per_launch": alg_bytes,
                      "stored_row_bytes": stored_row_bytes(args.store, args.metric, n_local),
                      "avg_launch_ms": scan_avg_ms,
                      "scan_alone": ({"kernel": scan_kernel, "avg_launch_ms": scan_alone_ms,
                                      "frac": alg_bytes / (scan_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "what": "the scan launch of a plain search (no gate workgroups behind the scan's), "
                                              "40 searches after the timed region"} if scan_alone_ms else None),
                      "launches_per_pass": launches, "launches_timed": len(scan_ms)}),
        "roofline_gate": {"bound": "mfma", "kernel": GATE_KERNELS[1 if os.environ.get("PRAG_PROBER_SHAPE") == "32" else 0], "achieved": gate_flops / (gate_avg_ms * 1e-3) / 1e12,
                          "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                          # (the rate a bare 16x16x32 MFMA loop sustains on this chip with 256 workgroups issuing - power:
                          #  profiles/r04a_mfma_shape_clock.txt - for reference beside the nominal dense peak)
                          "frac_of_sustained_mfma_rate_1940TF": gate_flops / (gate_avg_ms * 1e-3) / 1e12 / 1940.0,
                          "frac": gate_flops / (gate_avg_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF,
                          "avg_launch_ms": gate_avg_ms, "avg_launch_ms_back_to_back_alone": gate_alone_ms,
                          "timing": gate_timing,
                          "matrix_pipe_busy_frac_of_cu_busy": gate_mfma_busy, "matrix_pipe_busy_source": gate_mfma_note,
                          "hbm_GBs": (L * Bg * d_model * 2 + L * 1318914 * 2) / (gate_avg_ms * 1e-3) / 1e9},
    }

    # ---- variants: the other call shapes of SURVEY.md section 8d on the same corpus size (1 GPU) --------
    if world == 1 and not args.no_variants:
        variants = {}
        local.set_scan_workgroups(0)          # searches alone on the chip
        if not args.no_shard_variant and args.docs > SHARD_ROWS:
            try:
                variants[f"shard_{SHARD_ROWS}_q{SHARD_QUERIES}_gate{SHARD_GATE_ROWS}"] = \
                    shard_variant(torch, pra, ens, args.k, args.metric, ms_per_step / args.inner, adaptive=bool(args.adaptive))
            except Exception as e:
                variants[f"shard_{SHARD_ROWS}_q{SHARD_QUERIES}_gate{SHARD_GATE_ROWS}"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            qv = torch.from_numpy(synth_rows(7, 0, 1000, d_emb)).cuda()
            if args.store == "f16" and args.metric == "cos":
                # the stored fp16 rows scanned directly (no shadow), then the two-level search
                variants[f"f16_cos_k{args.k}_q{args.queries}_rows_scanned_directly"] = \
                    variant_record(torch, local, q, args.k, "f16", "cos", n_local, 0)
                for B in (1, 32, 1000):
                    variants[f"f16_cos_k{args.k}_q{B}"] = variant_record(torch, local, qv[:B], args.k, "f16", "cos", n_local, 0)
                variants[f"f16_cos_k{args.k}_q128"] = variant_record(torch, local, qv[:128], args.k, "f16", "cos", n_local, 0)
                for B in (1, 32, 128):
                    variants[f"f16_cos_k{args.k}_q{B}_shadow"] = variant_record(torch, local, qv[:B], args.k, "f16", "cos", n_local, 1)
                if n_local >= MM8_MIN_ROWS:     # > 128 queries with the shadow kept: int8 tiles first, fp16 tiles if a query fails
                    variants[f"f16_cos_k{args.k}_q1000_shadow"] = variant_record(torch, local, qv, args.k, "f16", "cos", n_local, 1)
                local.set_shadow(1 if args.shadow else 0)
            # the reference's literal call: IndexFlatL2 (float32 rows), one query, k = 5
            # (make_indexer.py:449-450, utils.py:378-380, exp_rag.py:432)
            ref_ix = pra.HipFlatIndex(d_emb, "l2", "f32", capacity=args.docs)
            ref_ix.add_synthetic(42, 0, args.docs)
            variants["f32_l2_k5_q1 (reference call)"] = variant_record(torch, ref_ix, qv[:1], 5, "f32", "l2", args.docs, 0)
            variants["f32_l2_k5_q32"] = variant_record(torch, ref_ix, qv[:32], 5, "f32", "l2", args.docs, 0)
            variants["f32_l2_k5_q1_shadow (reference call, two-level)"] = \
                variant_record(torch, ref_ix, qv[:1], 5, "f32", "l2", args.docs, 1)
            # the gate at the reference's call shape (one query: six probers, float32 states)
            e32 = pra.HipProberEnsemble(L, d_model, 2, weights="f32")
            for l, st in enumerate(states):
                e32.load_layer(l, st)
            # the reference-precision gate at the config-2 batch: float32 pooled states - what
            # exp_rag.py:385-387 hands the prober - through weights kept to ~22 bits (hi + lo fp16 terms, logits within
            # ~2e-6 of the fp32 module): a normalise-and-split pre-pass, then 3 MFMAs per product term
            try:
                x32 = torch.randn((L, args.gate_batch, d_model), device="cuda")
                e32.reserve(args.gate_batch)
                for _ in range(5):
                    e32.gate(x32, 0, 0.0)
                torch.cuda.synchronize()
                e32.profile(64)
                t1 = time.perf_counter()
                for _ in range(40):
                    e32.gate(x32, 0, 0.0)
                torch.cuda.synchronize()
                us32 = (time.perf_counter() - t1) / 40 * 1e6
                k32 = e32.profile_read()
                e32.profile(0)
                k32_us = float(np.mean(k32)) * 1e3 if k32 else None
                alg32 = L * args.gate_batch * d_model * 4 + L * 1318914 * 4 + L * args.gate_batch * 8
                rec32 = {"what": "HipProberEnsemble(weights='f32').gate on float32 states [6, %d, 2048]: prenorm_split_kernel "
                                 "(LayerNorm-0 in fp32, hi + lo fp16 split) + prober_fused_kernel (3 MFMAs per product) + "
                                 "gate_kernel, host-timed back to back" % args.gate_batch,
                         "us_per_call": us32, "decisions_per_s": args.gate_batch / us32 * 1e6,
                         "prober_fused_kernel_us": k32_us,
                         "algorithmic_bytes": alg32, "frac_of_8TBs_on_algorithmic_bytes": alg32 / us32 / 1e3 / HBM_PEAK_GBS,
                         "flops_reference": gate_flops, "mfma_flops_issued": 3.0 * gate_flops,
                         "frac_of_2.5PF_on_reference_flops": gate_flops / us32 / 1e6 / MFMA_F16_PEAK_TF,
                         "frac_of_2.5PF_on_issued_mfma_flops": 3.0 * gate_flops / us32 / 1e6 / MFMA_F16_PEAK_TF,
                         "over_f16_mode": us32 / (gate_alone_ms * 1e3) if gate_alone_ms else None}
                if args.measure_traffic:
                    pmc32, note32 = measure_gate_f32_pmc(args.gate_batch)
                    rec32["pmc"] = pmc32
                    rec32["pmc_source"] = note32
                variants["gate_f32w_f32x_b%d (reference precision at the config-2 batch)" % args.gate_batch] = rec32
                del x32
            except Exception as e:      # noqa: BLE001
                variants["gate_f32w_f32x (reference precision at the config-2 batch)"] = {"error": f"{type(e).__name__}: {e}"}
            x1 = torch.randn((L, 1, d_model), device="cuda")
            for _ in range(10):
                e32.gate(x1, 0, 0.0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(500):
                e32.gate(x1, 0, 0.0)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t1) / 500 * 1e6
            w_bytes = L * 1318914 * 4                # the small-batch path reads the folded weights as fp32 rows
            variants["gate_b1_f32 (reference call)"] = {
                "us_per_decision": us, "decisions_per_s": 1e6 / us,
                "what": "fused 6-prober gate, one pooled state, fp32-parity weights, host-timed back to back",
                "bound": "hbm", "algorithmic_bytes": w_bytes, "achieved": w_bytes / us / 1e3, "unit": "GB/s",
                "peak": HBM_PEAK_GBS, "frac": w_bytes / us / 1e3 / HBM_PEAK_GBS}
            # ... and as the retrieve-decide loop sees it (exp_rag.py:393, 406-415 end in a HOST branch): one decision
            # at a time, each one waited for - prag_gate_decide (one C call: kernels + the decision in host memory)
            # next to ens.gate + int(decision[0]) (what bench_e2e.py did in round 4)
            def _lat(fn, n=400):
                for _ in range(20):
                    fn()
                t_ = time.perf_counter()
                for _ in range(n):
                    fn()
                return (time.perf_counter() - t_) / n * 1e6
            us_decide = _lat(lambda: int(e32.decide(x1, 0, 0.0)[0]))
            us_gate_item = _lat(lambda: int(e32.gate(x1, 0, 0.0)[2][0]))
            variants["gate_b1_latency (reference call, decision on the host)"] = {
                "us_per_decision": us_decide, "us_per_decision_gate_then_item": us_gate_item,
                "kernel_us_back_to_back": us,
                "what": "ens.decide(x [6,1,2048] f32): gate kernels + decision written to pinned host memory + wait, per "
                        "call, host-timed with every call waited for; second figure: ens.gate(...) then int(decision[0])"}
            # ... and where the loop really calls it: after a Gemma-2B-shaped `generate` has emptied the caches, on the
            # device AND on the host.  `step`: the decode steps' pooling launches decide as they go
            # (round 6); `decide`: round 5's call after `generate` has returned
            if not args.no_gate_in_loop:
                try:
                    import bench_e2e
                    g_rec = bench_e2e.gate_in_loop(torch, pra, states, torch.device("cuda", dev_index))
                    g_rec["cold_us (512 MB read between calls, host warm)"] = bench_e2e.gate_cold(
                        torch, pra, states, torch.device("cuda", dev_index))
                    variants["gate_b1_in_loop (Gemma-2B-shaped generate between calls)"] = g_rec
                except Exception as e:      # noqa: BLE001 - e.g. no transformers on the box
                    variants["gate_b1_in_loop (Gemma-2B-shaped generate between calls)"] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            # the reference's retrieval call as the loop sees it: index.search(np.float32 [1,768], k=5) -> numpy
            # (utils.py:378-380, exp_rag.py:432-436), host buffers in and out, every call waited for
            lat_ix = ref_ix
            q1_np = qv[:1].cpu().numpy()
            lat_rec = {}
            for sh, name in ((0, "rows_scanned_directly"), (1, "two_level")):
                lat_ix.set_shadow(sh)
                lat_ix.prepare()
                lat_rec[name + "_us"] = _lat(lambda: lat_ix.search(q1_np, 5), n=60)
            lat_ix.close()
            lat_rec["what"] = ("index.search(np.float32 [1,768], 5) on IndexFlatL2-shaped float32 rows x %d: host-timed per "
                               "call, H2D of the query + kernels + D2H of (D, I) + wait" % args.docs)
            variants["search_b1_latency_host_io (reference call)"] = lat_rec
            # BASELINE config 3: 1k queries x 1M docs cosine top-10 (MFMA-tiled scan)
            c3 = pra.HipFlatIndex(d_emb, "cos", "f16", capacity=1_000_000)
            c3.add_synthetic(42, 0, 1_000_000)
            variants["C3_f16_cos_k10_q1000_x_1M"] = variant_record(torch, c3, qv, 10, "f16", "cos", 1_000_000, 0)
            c3.close()
            # a CLUSTERED corpus: 4 M rows around 4096 centres (cosine to the own centre ~0.9), queries
            # near centres - thousands of rows sit inside the shadow's error band of the k-th score, the case
            # where candidate regions can overflow into the exact scan; fallbacks are part of the record
            variants["clustered_f16_cos_k10_q64_x_4M"] = clustered_variant(torch, pra, d_emb, args.k)
            variants["clustered_interleaved_f16_cos_k10_q64_x_4M"] = clustered_variant(torch, pra, d_emb, args.k, interleaved=True)
            # embedding-shaped corpora (common mean, power-law spectrum, outlier coordinates), next to their iid
            # counterparts above: 4 M rows cosine / L2, and the headline size
            emb = {"what": "rows = mu + U diag(lambda) z, ||mu|| = 0.8 ||x||, lambda_j ~ j^-0.5, 6 outlier coordinates at "
                           "10-30 x the median |x_j|, un-normalised; 64 queries of the same distribution, top-%d" % args.k}
            emb["cos_4M"] = embedding_variant(torch, pra, d_emb, args.k, 4_194_304, "cos", "f16")
            emb["l2_4M"] = embedding_variant(torch, pra, d_emb, args.k, 4_194_304, "l2", "f16")
            emb["cos_21M"] = embedding_variant(torch, pra, d_emb, args.k, args.docs, "cos", "f16")
            variants["embedding_like_f16_k10_q64"] = emb
        except Exception as e:                # a variant must never take the headline down with it
            variants["error"] = f"{type(e).__name__}: {e}"
        out["variants"] = variants
        # the driver's record keeps `config` and `roofline` whole: the second half of the metric and the
        # numbers the round is judged on go there as well
        sv = variants.get(f"shard_{SHARD_ROWS}_q{SHARD_QUERIES}_gate{SHARD_GATE_ROWS}") or {}
        c3v = variants.get("C3_f16_cos_k10_q1000_x_1M") or {}
        b1 = variants.get("gate_b1_f32 (reference call)") or {}
        b1l = variants.get("gate_b1_latency (reference call, decision on the host)") or {}
        s1l = variants.get("search_b1_latency_host_io (reference call)") or {}
        q128 = variants.get(f"f16_cos_k{args.k}_q128_shadow") or {}
        gil = variants.get("gate_b1_in_loop (Gemma-2B-shaped generate between calls)") or {}
        g32 = variants.get("gate_f32w_f32x_b%d (reference precision at the config-2 batch)" % args.gate_batch) or {}
        emb = variants.get("embedding_like_f16_k10_q64") or {}
        out["config"].update({
            "shard_pass_ms": sv.get("pass_ms"), "shard_pass_mode": sv.get("pass_mode"),
            "shard_pass_ms_by_mode": sv.get("pass_ms_by_mode"),
            "shard_ids_identical_to_direct_scan": sv.get("ids_identical_to_direct_scan_of_the_rows"),
            "predicted_strong_scaling_eff": sv.get("predicted_strong_scaling_eff"),
            "c3_ms": c3v.get("ms_per_search"), "c3_frac_of_2.5PF": c3v.get("frac"),
            "gate_b1_us": b1.get("us_per_decision"), "gate_b1_latency_us": b1l.get("us_per_decision"),
            "gate_f32w_f32x_b4096_us": g32.get("us_per_call"), "gate_f32w_f32x_decisions_per_s": g32.get("decisions_per_s"),
            "gate_f32w_f32x_frac_hbm": g32.get("frac_of_8TBs_on_algorithmic_bytes"),
            "gate_f32w_f32x_frac_mfma_issued": g32.get("frac_of_2.5PF_on_issued_mfma_flops"),
            "gate_f32w_f32x_over_f16": g32.get("over_f16_mode"),
            "clustered_interleaved_two_level_ms": ((variants.get("clustered_interleaved_f16_cos_k10_q64_x_4M") or {}).get("two_level") or {}).get("ms_per_search"),
            "clustered_interleaved_fallbacks": ((variants.get("clustered_interleaved_f16_cos_k10_q64_x_4M") or {}).get("two_level") or {}).get("exact_fallbacks_last_search"),
            "gate_b1_cold_us": gil.get("cold_us (512 MB read between calls, host warm)"),
            "gate_b1_in_loop_us": gil.get("step_gate_us"), "gate_b1_in_loop_r5_call_us": gil.get("decide_gate_us"),
            "search_b1_latency_us": s1l.get("two_level_us"), "q128_shadow_ms": q128.get("ms_per_search"),
            "embedding_like": {k_: emb.get(k_) for k_ in ("cos_4M", "l2_4M", "cos_21M")} if emb else None})
        out["roofline"]["shard_scan8_frac"] = sv.get("scan8_frac_of_8TBs")
        out["roofline"]["shard_scan8_ms"] = sv.get("scan8_kernel_ms_in_pass")
    else:
        out["variants"] = None
    out["config"]["probe_decisions_per_s"] = out["probe_decisions_per_s"]
    out["config"]["exchange"] = index.exchange

    if world == 1 and not args.no_cpu_baseline:
        q_c1 = synth_rows(7, 0, 128, d_emb)
        x_c1 = synth_rows(42, 0, 10_000, d_emb)
        out["cpu_baseline"] = cpu_baseline(args, states, q_c1, x_c1)
    else:
        out["cpu_baseline"] = None
    emit(out, real_stdout)
    if world > 1:
        index.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:      # a rank that dies says who it was before the launcher tears the job down
        print(f"bench.py: rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} "
              f"(local rank {os.environ.get('LOCAL_RANK', '0')}, pid {os.getpid()}, backend "
              f"{os.environ.get('PRAG_BENCH_BACKEND', 'nccl')}) failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        raise
