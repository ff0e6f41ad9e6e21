"""GPU: RCCL itself (backend "nccl") with the one rank a 1-GPU box allows - initialisation as in bench.py and every
collective of the row-sharded path in its real dtype and shape (packed uint8 all-gather of the local top-k lists,
int64 shard sizes, float64 MAX, all_gather_object, barrier).  The 2-rank tests run over gloo because RCCL refuses
two ranks on one device; this one makes sure the RCCL calls themselves are well-formed."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_collectives_of_the_sharded_path_with_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_one_rank_worker.py")], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RCCL_ONE_RANK_OK" in out.stdout
