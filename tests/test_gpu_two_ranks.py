"""The multi-rank paths on real kernels: two ranks sharing the one GPU of the test box (tests/dist_two_rank_check.py).
The CPU-only world-size-2 tests (test_host_logic.py, test_training_host.py) cover the host logic; RCCL itself needs
one GPU per rank and is exercised by `bench.py --gpus N` on the multi-GPU node."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_share_one_gpu(gpu_device):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    with socket.socket() as sock:  # a port nobody holds right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dist_two_rank_check.py")]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "TWO-RANK OK" in run.stdout
