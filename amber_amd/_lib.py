"""ctypes binding of libamber_mpn.so (include/amber_mpn.h).

There is no CPU path: if the library is missing or cannot be loaded, importing the compute modules
fails loudly.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C amber_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMBER_MPN_LIB") or os.path.join(_HERE, "csrc", "libamber_mpn.so")  # override: kernel experiments

MAX_MLP_LAYERS = 4
TILE = 32
NUM_KERNEL_CLASSES = 6
CSR_STATIC_SCRATCH_BYTES = 16 << 20  # AMBER_MPN_CSR_STATIC_SCRATCH_BYTES

ACT = {"relu": 0, "leakyrelu": 1, "tanh": 2, "elu": 3, "swish": 4, "silu": 4, "mish": 5, "gelu": 6}
MODE = {None: 0, "none": 0, "inner": 1, "outer": 2}
NORM = {None: 0, "none": 0, "layer": 1, "layer_norm": 1, "batch": 2, "batch_norm": 2}


class AmberMpnDesc(C.Structure):
    _fields_ = [
        ("latent", C.c_int32),
        ("num_steps", C.c_int32),
        ("num_step_repeats", C.c_int32),
        ("mlp_layers", C.c_int32),
        ("activation", C.c_int32),
        ("residual", C.c_int32),
        ("layer_norm", C.c_int32),
        ("in_node_features", C.c_int32),
        ("in_edge_features", C.c_int32),
        ("embed_layers", C.c_int32),
        ("embed_activation", C.c_int32),
        ("has_head", C.c_int32),
        ("decoder_layers", C.c_int32),
        ("decoder_activation", C.c_int32),
        ("decoder_norm", C.c_int32),
        ("precision", C.c_int32),
        ("ln_eps", C.c_float),
    ]


TRANSFORM = {None: 0, False: 0, "none": 0, "null": 0, "softplus": 1, "exp": 2, "exponential": 2}


class AmberGlueDesc(C.Structure):
    _fields_ = [
        ("normalize_inputs", C.c_int32),
        ("denormalize_predictions", C.c_int32),
        ("transform", C.c_int32),
        ("predict_residual", C.c_int32),
        ("apply_transform", C.c_int32),
        ("clip", C.c_int32),
        ("clip_min", C.c_float),
        ("clip_max", C.c_float),
        ("scale0", C.c_float),
        ("scale1", C.c_float),
        ("input_clip", C.c_double),
        ("epsilon", C.c_double),
    ]


class AmberGraphDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "add_self_edges", "distance_vector", "euclidean_distance", "edge_curvature",
                                         "reserved")] + [("edge_capacity", C.c_int64), ("edge_offset", C.c_int64)]


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_SZ = C.c_size_t
_DESC = C.POINTER(AmberMpnDesc)
_GLUE = C.POINTER(AmberGlueDesc)
_F64 = C.c_double

# name -> (restype, argtypes): every symbol include/amber_mpn.h declares
SIGNATURES = {
    "amber_mpn_abi_version": (C.c_int, []),
    "amber_mpn_last_error": (C.c_char_p, []),
    "amber_mpn_param_floats": (_I64, [_DESC]),
    "amber_mpn_packed_floats": (_I64, [_DESC]),
    "amber_mpn_pack_weights": (C.c_int, [_DESC, _P, _P, _P]),
    "amber_mpn_csr_workspace_bytes": (_SZ, [_I64, _I64]),
    "amber_mpn_csr_build": (C.c_int, [_P, _I64, _I64, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "amber_mpn_csr_build_static": (C.c_int, [_P, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "amber_mpn_segment_mean": (C.c_int, [_P, _P, _I64, _I64, _I32, _P, _P]),
    "amber_mpn_permute_rows": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P]),
    "amber_mpn_workspace_bytes": (_SZ, [_DESC, _I64, _I64]),
    "amber_mpn_embed": (C.c_int, [_DESC, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "amber_mpn_block": (C.c_int, [_DESC, _P, _P, _I32, _I64, _I64, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "amber_mpn_project": (C.c_int, [_DESC, _P, _P, _I32, _I64, _P, _P, _P, _P]),
    "amber_mpn_head": (C.c_int, [_DESC, _P, _P, _I64, _P, _P, _P, _P]),
    "amber_mpn_forward": (C.c_int, [_DESC, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "amber_mpn_forward_timed": (C.c_int, [_DESC, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P,
                                          C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "amber_mpn_tape_bytes": (_SZ, [_DESC, _I64, _I64]),
    "amber_mpn_forward_train": (C.c_int, [_DESC, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _P]),
    "amber_mpn_backward_workspace_bytes": (_SZ, [_DESC, _I64, _I64]),
    "amber_mpn_backward": (C.c_int, [_DESC, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _P,
                                     _SZ, _P]),
    "amber_mpn_normalize_rows": (C.c_int, [_P, _I64, _I32, _P, _P, _F64, _F64, _P, _P]),
    "amber_mpn_postprocess": (C.c_int, [_GLUE, _P, _P, _P, _P, _I64, _P, _P]),
    "amber_mpn_postprocess_backward": (C.c_int, [_GLUE, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "amber_mpn_inverse_transform": (C.c_int, [_I32, _P, _P, _I64, _P, _P]),
    "amber_mpn_graph_edge_features": (_I32, [C.POINTER(AmberGraphDesc)]),
    "amber_mpn_build_graph": (C.c_int, [C.POINTER(AmberGraphDesc), _I32, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "amber_mpn_build_inter_edges": (C.c_int, [C.POINTER(AmberGraphDesc), _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "amber_mpn_project_workspace_bytes": (_SZ, [_I32, _I64, _I64]),
    "amber_mpn_project_elements_to_vertices": (C.c_int, [_P, _I32, _I64, _I64, _P, _P, _P, _P, _P, _SZ, _P]),
    "amber_mpn_estimate_workspace_bytes": (_SZ, [_I64, _I64]),
    "amber_mpn_estimate_num_elements": (C.c_int, [_P, _I32, _I64, _I64, _P, _P, _I32, _I32, _P, _I64, _P, _F64, _P, _P, _SZ, _P]),
    "amber_mpn_predict_workspace_bytes": (_SZ, [_DESC, _I64, _I64]),
    "amber_mpn_predict": (C.c_int, [_DESC, _GLUE, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                    _P, _SZ, _P]),
}

# the sources the split-precision forward kernels are compiled from (what the PMC traffic figure measures)
_HASHED_SOURCES = ("mpn_f2.hip", "mpn_tile.h")


def csrc_hash() -> str:
    """Content hash of the sources of the split-precision forward kernels: stamps counter figures taken from one build
    (profiles/traffic.json) so that bench.py does not report them for another.  Other translation units (backward, CSR,
    glue ...) do not enter: a change there does not touch the measured kernel."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    for name in _HASHED_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises RuntimeError when it is missing: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be loaded FIRST: it bundles its own libamdhip64.so (SONAME libamdhip64.so.7), which then also
    # satisfies this library's dependency, so both share ONE HIP runtime (device pointers, streams).  Loaded
    # the other way round, the process ends up with two runtimes and every call here fails with "no device".
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"amber_amd: HIP extension not built ({LIB_PATH} missing). Run `make -C amber_amd/csrc` "
            f"(or __graft_entry__.build()). There is no CPU fallback."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise RuntimeError(f"amber_amd: cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.amber_mpn_abi_version() != 2:
        raise RuntimeError("amber_amd: libamber_mpn.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().amber_mpn_last_error()
        raise RuntimeError(f"amber_amd: {what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr(device) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream
