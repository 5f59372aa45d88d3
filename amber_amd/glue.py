"""Host wrappers of the pre / post glue kernels (include/amber_mpn.h: amber_mpn_normalize_rows, amber_mpn_postprocess,
amber_mpn_postprocess_backward, amber_mpn_inverse_transform).  GPU tensors only: there is no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .csr import _require_cuda


def make_glue_desc(*, normalize_inputs: bool = False, denormalize_predictions: bool = False, transform=None,
                   predict_residual: bool = False, apply_transform: bool = True, clip_min: Optional[float] = None,
                   clip_max: Optional[float] = None, scale0: float = 1.0, scale1: float = 1.0, input_clip: float = 10.0,
                   epsilon: float = 1.0e-6) -> _lib.AmberGlueDesc:
    key = transform.lower() if isinstance(transform, str) else transform
    if key not in _lib.TRANSFORM:
        raise ValueError(f"Unsupported transform type: transform_name={transform!r}")
    return _lib.AmberGlueDesc(
        normalize_inputs=int(bool(normalize_inputs)), denormalize_predictions=int(bool(denormalize_predictions)),
        transform=_lib.TRANSFORM[key], predict_residual=int(bool(predict_residual)), apply_transform=int(bool(apply_transform)),
        clip=(1 if clip_min is not None else 0) | (2 if clip_max is not None else 0),
        clip_min=0.0 if clip_min is None else float(clip_min), clip_max=0.0 if clip_max is None else float(clip_max),
        scale0=float(scale0), scale1=float(scale1), input_clip=float(input_clip), epsilon=float(epsilon))


def _stats(t: torch.Tensor, device, what: str) -> torch.Tensor:
    """A float64 statistics buffer as a contiguous device tensor (its pointer goes to the kernel)."""
    if t.dtype != torch.float64:
        raise TypeError(f"amber_amd: {what} must be float64 (TorchRunningMeanStd buffers), got {t.dtype}")
    return t.to(device).contiguous()


def normalize_rows(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, epsilon: float, clip: float) -> torch.Tensor:
    """RunningNormalizer._normalize on rows of features: float32(clip((x - mean) / sqrt(var + eps), +-clip)), bit-identical
    to the reference's float64 arithmetic."""
    _require_cuda(x, "input_tensor")
    xs = x.to(torch.float32).contiguous()
    width = int(xs.shape[-1]) if xs.ndim > 0 else 1
    if mean.numel() != width or var.numel() != width:
        raise ValueError(f"amber_amd: {width} features, but statistics of {mean.numel()}")
    out = torch.empty_like(xs)
    rows = xs.numel() // max(width, 1)
    m, v = _stats(mean, x.device, "mean"), _stats(var, x.device, "var")
    with torch.cuda.device(x.device):
        rc = _lib.load().amber_mpn_normalize_rows(xs.data_ptr(), rows, width, m.data_ptr(), v.data_ptr(), float(epsilon), float(clip),
                                                  out.data_ptr(), _lib.stream_ptr(x.device))
    _lib.check(rc, "normalize_rows")
    return out


class _Post(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, desc, baseline, mean, var):
        lib = _lib.load()
        out = torch.empty_like(pred)
        with torch.cuda.device(pred.device):
            rc = lib.amber_mpn_postprocess(C.byref(desc), pred.data_ptr(), _lib.ptr(baseline), _lib.ptr(mean), _lib.ptr(var), pred.numel(),
                                           out.data_ptr(), _lib.stream_ptr(pred.device))
        _lib.check(rc, "postprocess")
        ctx.desc, ctx.baseline, ctx.mean, ctx.var = desc, baseline, mean, var
        ctx.save_for_backward(pred)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (pred,) = ctx.saved_tensors
        lib = _lib.load()
        g = grad_out.to(torch.float32).contiguous()
        grad_pred = torch.empty_like(pred)
        with torch.cuda.device(pred.device):
            rc = lib.amber_mpn_postprocess_backward(C.byref(ctx.desc), pred.data_ptr(), _lib.ptr(ctx.baseline), _lib.ptr(ctx.mean),
                                                    _lib.ptr(ctx.var), g.data_ptr(), pred.numel(), grad_pred.data_ptr(),
                                                    _lib.stream_ptr(pred.device))
        _lib.check(rc, "postprocess_backward")
        return grad_pred, None, None, None, None


def postprocess(pred: torch.Tensor, *, transform=None, predict_residual: bool = False, baseline: Optional[torch.Tensor] = None,
                apply_transform: bool = True, mean: Optional[torch.Tensor] = None, var: Optional[torch.Tensor] = None,
                epsilon: float = 1.0e-6, clip_min: Optional[float] = None, clip_max: Optional[float] = None, scale0: float = 1.0,
                scale1: float = 1.0) -> torch.Tensor:
    """denormalise (if mean / var are given) -> + transform^-1(baseline) (residual) -> transform -> clip -> damping, one
    kernel; differentiable with respect to `pred` (without clip / damping)."""
    _require_cuda(pred, "predictions")
    shape = pred.shape
    p = pred.to(torch.float32).contiguous().view(-1)
    desc = make_glue_desc(denormalize_predictions=mean is not None, transform=transform, predict_residual=predict_residual,
                          apply_transform=apply_transform, clip_min=clip_min, clip_max=clip_max, scale0=scale0, scale1=scale1,
                          epsilon=epsilon)
    b = None
    if predict_residual and apply_transform:
        if baseline is None:
            raise AssertionError("Baseline must be provided for residual prediction.")
        b = baseline.to(device=pred.device, dtype=torch.float32).contiguous().view(-1)
        if b.numel() != p.numel():
            raise ValueError(f"amber_amd: {p.numel()} predictions but a baseline of {b.numel()}")
    m = _stats(mean, pred.device, "mean") if mean is not None else None
    v = _stats(var, pred.device, "var") if var is not None else None
    if m is not None and (m.numel() != 1 or v is None or v.numel() != 1):
        raise ValueError("amber_amd: the prediction normaliser must hold one statistic (num_predictions = 1)")
    return _Post.apply(p, desc, b, m, v).view(shape)


def denormalize(pred: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, epsilon: float) -> torch.Tensor:
    """RunningNormalizer._denormalize: float32(pred * sqrt(var + eps) + mean) in float64 arithmetic."""
    return postprocess(pred, transform=None, apply_transform=False, mean=mean, var=var, epsilon=epsilon)


def inverse_transform(t: torch.Tensor, transform, baseline: Optional[torch.Tensor] = None) -> torch.Tensor:
    """transform^-1(t) (- transform^-1(baseline) for residual prediction), one kernel."""
    _require_cuda(t, "labels")
    key = transform.lower() if isinstance(transform, str) else transform
    if key not in _lib.TRANSFORM:
        raise ValueError(f"Unsupported transform type: transform_name={transform!r}")
    src = t.to(torch.float32).contiguous()
    b = None
    if baseline is not None:
        b = baseline.to(device=t.device, dtype=torch.float32).contiguous()
        if b.numel() != src.numel():
            raise ValueError(f"amber_amd: {src.numel()} labels but a baseline of {b.numel()}")
    out = torch.empty_like(src)
    with torch.cuda.device(t.device):
        rc = _lib.load().amber_mpn_inverse_transform(_lib.TRANSFORM[key], src.data_ptr(), _lib.ptr(b), src.numel(), out.data_ptr(),
                                                     _lib.stream_ptr(t.device))
    _lib.check(rc, "inverse_transform")
    return out
