"""`python bench.py --gpus N` must start its own one-rank-per-GPU launch: a fresh child process,
created before torch is imported or the GPU is touched in the parent, never an exec (SCALE runs
call `python bench.py --gpus N` directly when no launcher wraps it)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import json, os, sys
sys.path.insert(0, {root!r})
os.environ.pop("WORLD_SIZE", None)
import bench
calls = []
class _Done:
    returncode = 7
    pid = 0
    def wait(self, timeout=None):
        return 7
def fake_run(cmd, env=None, **kw):
    calls.append({{"cmd": cmd, "torch_loaded": "torch" in sys.modules,
                   "ipc": (env or {{}}).get("HSA_ENABLE_IPC_MODE_LEGACY"), "new_session": kw.get("start_new_session")}})
    return _Done()
bench.subprocess.run = fake_run
bench.subprocess.Popen = fake_run
for name in ("execv", "execve", "execvp", "execvpe", "execl", "execle", "execlp"):
    setattr(os, name, lambda *a, **k: (_ for _ in ()).throw(AssertionError("exec used")))
try:
    bench.main(["--gpus", "4", "--steps", "2", "--warmup", "1"])
    code = None
except SystemExit as e:
    code = e.code
print(json.dumps({{"calls": calls, "code": code, "torch_loaded_after": "torch" in sys.modules}}))
"""


def _run_probe(extra_env=None):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, "-c", _PROBE.format(root=ROOT)], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    import json
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_gpus_n_spawns_torchrun_child_before_torch_is_imported():
    rec = _run_probe()
    assert len(rec["calls"]) == 1
    call = rec["calls"][0]
    cmd = call["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], os.path.join(ROOT, "bench.py"))
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]     # the flags travel unchanged
    assert call["torch_loaded"] is False and rec["torch_loaded_after"] is False   # no GPU initialised
    assert call["ipc"] == "0"
    assert call["new_session"] is True                                        # killable as a group on timeout
    assert rec["code"] == 7                                                   # the child's exit code


def test_a_hung_launch_is_killed_as_a_group_and_exits_nonzero(monkeypatch, tmp_path):
    """A rendezvous or collective that never completes must not hang the caller: after --launch-timeout the child
    AND its descendants (the ranks) are killed and the exit code is 124; the parent is never replaced."""
    import signal
    import time
    import bench
    pidfile = tmp_path / "grandchild.pid"
    child = ("import os, subprocess, sys, time\n"
             "p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"
             f"open({str(pidfile)!r}, 'w').write(str(p.pid))\n"
             "time.sleep(600)\n")
    monkeypatch.setattr(bench, "launch_command", lambda n, argv, port=None: [sys.executable, "-c", child])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delitem(sys.modules, "torch", raising=False)
    args = bench.parse(["--gpus", "2", "--launch-timeout", "3"])
    t0 = time.time()
    rc = bench.self_launch(args, ["--gpus", "2"])
    assert rc == 124 and time.time() - t0 < 60
    gpid = int(pidfile.read_text())
    for _ in range(50):                       # the grandchild went down with the group
        try:
            os.kill(gpid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        os.kill(gpid, signal.SIGKILL)
        raise AssertionError("the rank processes survived the launcher's timeout")


def test_a_rank_does_not_launch_again():
    import bench
    args = bench.parse(["--gpus", "4"])
    old = os.environ.get("WORLD_SIZE")
    os.environ["WORLD_SIZE"] = "4"
    try:
        assert bench.self_launch(args, ["--gpus", "4"]) is None
    finally:
        if old is None:
            os.environ.pop("WORLD_SIZE")
        else:
            os.environ["WORLD_SIZE"] = old
    assert bench.self_launch(bench.parse(["--gpus", "1"]), []) is None


def test_nccl_with_too_few_gpus_exits_nonzero():
    """A rank under the RCCL backend on a box with fewer GPUs than ranks must fail, not oversubscribe."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    env.pop("PRAG_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "RCCL ranks need" in out.stderr


def test_traffic_is_never_measured_from_inside_a_profiled_run(monkeypatch):
    """roofline.traffic comes from child `rocprofv3 --pmc` passes; a run that is itself under a profiler (the
    profiler's environment is inherited) or a box without rocprofv3 must fall back to the committed summary
    without starting anything."""
    import bench
    started = []
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: started.append(a))
    monkeypatch.setenv("ROCPROF_KERNEL_TRACE", "1")
    val, note = bench.measure_traffic(1000, "f16", "cos", 64, 1, "scan8_kernel")
    assert val is None and isinstance(note, str) and not started
    monkeypatch.delenv("ROCPROF_KERNEL_TRACE")
    monkeypatch.setenv("PATH", "/nonexistent")
    val, note = bench.measure_traffic(1000, "f16", "cos", 64, 1, "scan8_kernel")
    assert val is None and "rocprofv3" in note and not started


# ---- the line the driver parses (a 20.7 KB line left BENCH_r05.parsed = null) -----------------------
def _strict_loads(line):
    import json

    def _no_constants(name):
        raise AssertionError(f"{name} is not JSON")
    return json.loads(line, parse_constant=_no_constants)


def _full_record(n_ranks=1, source="r05f"):
    """Round 5's full single-GPU record (profiles/r05f_bench.json, 20.7 KB) or round 6's (the BENCH_DETAIL line of
    profiles/r06last_bench.json: more variants, more config fields), optionally dressed up as an N-rank run with every
    optional block present and over-long strings - the worst case the compact line has to survive."""
    import json
    if source == "r05f":
        full = json.load(open(os.path.join(ROOT, "profiles", "r05f_bench.json")))
    else:
        first = open(os.path.join(ROOT, "profiles", "r06last_bench.json")).read().splitlines()[0]
        assert first.startswith("BENCH_DETAIL ")
        full = json.loads(first[len("BENCH_DETAIL "):])
    if n_ranks > 1:
        full["n_gpus"] = full["rccl_ranks"] = n_ranks
        full["per_rank"] = [{"rank": r, "device": r, "rows": 2_625_000, "gate_rows": 512, "scan_ms": 0.38689423620700836,
                             "gate_ms": None, "allgather_us": 31.41592653589793, "timed_region_s": 1.0512253229971975,
                             "exact_fallbacks_last_search": 0} for r in range(n_ranks)]
        full["c_exchange_probe"] = {"ok": True, "ms_per_pass": 0.4512345678, "allgather_us": 27.123456789, "passes": 100}
        full["exchange_note"] = "x" * 900
        full["cpu_baseline"]["sample"] = "y" * 900
        full["roofline"]["nan_field"] = float("nan")
        full["config"]["ms_per_pass"] = float("inf")
    return full


def test_the_last_line_is_compact_strict_json_with_the_contract_keys():
    import bench
    for n, source in ((1, "r05f"), (8, "r05f"), (1, "r06last"), (8, "r06last")):
        full = _full_record(n, source)
        rec, line = bench.compact_record(full)
        assert len(line) < bench.COMPACT_LIMIT == 6000 and "\n" not in line
        got = _strict_loads(line)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in got, key
        assert got["value"] == full["value"] and got["ms_per_step"] == full["ms_per_step"]
        assert "workload" in got["config"] and "model" not in got["config"]
        assert not any(isinstance(v, (dict, list)) for v in got["config"].values())      # scalars only
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
            assert key in got["roofline"], key
        assert got["roofline"]["gate"]["kernel"] and "roofline_gate" not in got            # one copy of the gate block
        assert "definition" not in got["roofline"] and "variants" not in got
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in got["cpu_baseline"], key
        assert len(got["cpu_baseline"]["sample"]) <= 100 and len(got["dtype"]) <= 24
        if n > 1:
            assert len(got["per_rank"]) == n and got["per_rank"][0][0] == 2_625_000


def test_emit_prints_the_detail_first_and_the_compact_line_last(tmp_path, monkeypatch):
    import io
    import json
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    bench.emit(_full_record(8), buf)
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2 and lines[0].startswith("BENCH_DETAIL {")
    assert lines[-1].startswith("{") and len(lines[-1]) < bench.COMPACT_LIMIT
    last = _strict_loads(lines[-1])
    detail = _strict_loads(lines[0][len("BENCH_DETAIL "):])
    assert "variants" in detail and detail["value"] == last["value"]
    assert json.load(open(tmp_path / "gpurun_out" / "bench_detail.json"))["value"] == last["value"]


def test_library_chatter_cannot_reach_stdout():
    """After quiet_stdout() anything a library writes to fd 1 lands on stderr; only the saved stream is stdout."""
    code = ("import os, sys; sys.path.insert(0, %r); import bench\n"
            "real = bench.quiet_stdout()\n"
            "os.write(1, b'[Gloo] Rank 0 is connected\\n'); print('python chatter')\n"
            "real.write('{\"ok\": true}\\n'); real.flush()\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout == '{"ok": true}\n'
    assert "[Gloo]" in out.stderr and "python chatter" in out.stderr
