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


def test_bench_starts_its_ranks_itself_on_one_gpu(gpu_device):
    """`python bench.py --gpus 2` with no launcher around it (how the driver runs the bench): the parent spawns the two ranks,
    which here share the test box's GPU over gloo (`--rehearse-on-one-gpu`); rank 0's line must describe a two-rank job."""
    import json

    repo = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--graphs", "6",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=170, cwd=repo)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["scaling"] == "strong"
    assert len(line["ranks"]["edges"]) == 2 and min(line["ranks"]["edges"]) > 0
    assert "partition_of_the_one_batch" in line["ranks"] and line["value"] > 0
