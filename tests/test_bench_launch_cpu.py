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
