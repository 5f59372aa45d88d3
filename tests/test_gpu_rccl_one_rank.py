"""RCCL itself on the test box's one GPU: a one-rank process group (backend "nccl" = RCCL on ROCm) running the collectives
of the multi-rank paths - gradient all-reduce, barrier, all_gather - on device tensors, and the sharded forward through the
group.  No xGMI traffic with one rank; what it pins is that RCCL starts in this image / environment and that the calls
the N > 1 bench and the training step make are well-formed."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_one_rank_rccl_group_runs_the_collectives(gpu_device):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_one_rank.py"), str(port)], capture_output=True, text=True,
                       timeout=240, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
