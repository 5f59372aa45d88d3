"""Per-parameter gradient errors of the training path on golden fixtures (developer diagnostic, GPU box).
usage: python tools/diag_train_fixture.py <fixture> [precision]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_golden, load_golden_grads, rel_err
from amber_amd import engine
from test_gpu_training import _model, _batch

name = sys.argv[1]
engine.default_precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda:0")
cfg, params, d = load_golden(name)
grad_out, want = load_golden_grads(name)
model = _model(cfg, params, d, dev).eval()
out = model(_batch(d, dev))
print("forward err", rel_err(out.detach().cpu(), d["out_sizing"]))
(out * grad_out.to(dev)).sum().backward()
for n, p in model.named_parameters():
    print(f"{rel_err(p.grad.cpu(), want[n]):.3e}  {float(want[n].abs().max()):.3e}  {n}")
from test_gpu_training import _oracle_grads_with_kinks, _check_grads_kink_aware
_, w64, flips = _oracle_grads_with_kinks(cfg, params, d["x"], d["edge_attr"], d["edge_index"], d["mask_output"], grad_out)
print("units near the kink:", len(flips))
worst, coef = _check_grads_kink_aware(model, {k: v.double() for k, v in want.items()}, flips, lo=-1.0)
print("after accounting for them: worst", worst, "coefficients", coef.tolist())
