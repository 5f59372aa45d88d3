"""`python bench.py --gpus N` without a launcher (how the driver runs bench.py) must start the N ranks itself, and keep
working under torch.distributed.run.  CPU only: the dry-run hook joins a gloo group and makes no GPU call."""
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    return env


def _json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout  # rank 0 prints ONE JSON line, the other ranks none
    return json.loads(lines[0])


def test_self_launch_starts_the_ranks_and_they_rendezvous():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-rendezvous"], env=_clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["world_size"] == 2 and line["ranks_seen"] == [0, 1] and line["self_launched"] is True


def test_launcher_form_still_works():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29611", BENCH, "--gpus", "2", "--dry-run-rendezvous"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["world_size"] == 2 and line["ranks_seen"] == [0, 1] and line["self_launched"] is False


def test_a_failing_rank_fails_the_job():
    """A real failure on a CPU box: the ranks find no GPU -> the parent exits non-zero (and prints no JSON line) instead
    of hanging or reporting success."""
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs: there the real bench would start")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no GPU" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
