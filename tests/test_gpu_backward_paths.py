"""The three backward paths of a split-precision model give the same gradients: the fused kernels (default at latent <= 64),
the chain + weight-gradient kernels on bf16x3 operands (AMBER_MPN_NO_FUSED_BACKWARD), and the exact fp32 kernels
(AMBER_MPN_EXACT_BACKWARD).  The library reads these switches once per process, so every path runs in a process of its
own (tests/_bwd_path_worker.py) on the same seeded model and batch.  Each path is held to the oracle elsewhere
(tests/test_gpu_training.py runs the default path of every configuration); this pins that the switches select what they
say and that the paths agree far inside the 1e-4 gradient bar."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, tag, latent, **env):
    out = str(tmp_path / f"{tag}.npz")
    clean = {k: v for k, v in os.environ.items() if k not in ("AMBER_MPN_NO_FUSED_BACKWARD", "AMBER_MPN_EXACT_BACKWARD")}
    r = subprocess.run([sys.executable, os.path.join(HERE, "_bwd_path_worker.py"), out, str(latent)], capture_output=True, text=True,
                       timeout=240, env=dict(clean, **env))
    assert r.returncode == 0 and "BWD_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


@pytest.mark.parametrize("latent", [32, 64])
def test_fused_unfused_and_exact_backward_agree(gpu_device, tmp_path, latent):
    fused = _run(tmp_path, "fused", latent)
    unfused = _run(tmp_path, "unfused", latent, AMBER_MPN_NO_FUSED_BACKWARD="1")
    exact = _run(tmp_path, "exact", latent, AMBER_MPN_EXACT_BACKWARD="1")
    assert set(fused) == set(unfused) == set(exact) and len(fused) > 20
    np.testing.assert_array_equal(fused["pred"], unfused["pred"])  # the forward is the same kernel in all three
    np.testing.assert_array_equal(fused["pred"], exact["pred"])
    differ = 0
    for name, want in exact.items():
        if name == "pred":
            continue
        scale = max(float(np.abs(want).max()), 1e-30)
        for got in (fused[name], unfused[name]):
            assert float(np.abs(got - want).max()) <= 5e-6 * scale, name
        differ += int(not np.array_equal(fused[name], unfused[name])) + int(not np.array_equal(unfused[name], want))
    assert differ > 0  # the switches did select different arithmetic (bf16x3 / fused summation order / exact)
