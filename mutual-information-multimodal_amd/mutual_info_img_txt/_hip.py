"""ctypes binding of libmi_critic_hip.so (the C ABI declared in include/mi_critic.h).

PyTorch is plumbing here: it owns device memory and the stream; every compute call goes through the C ABI.
There is no CPU fallback: a missing library or a CPU tensor is an error.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("MI_CRITIC_LIB", os.path.join(_PKG_ROOT, "lib", "libmi_critic_hip.so"))

MI_DV, MI_INFONCE = 0, 1
MI_PREC_F32, MI_PREC_BF16, MI_PREC_BF16X3, MI_PREC_FP8, MI_PREC_F16, MI_PREC_F16X3 = 0, 1, 2, 3, 4, 5
ESTIMATORS = {"dv": MI_DV, "infonce": MI_INFONCE}
PRECISIONS = {"f32": MI_PREC_F32, "fp32": MI_PREC_F32, "float32": MI_PREC_F32, "f32_exact": MI_PREC_F32,
              "bf16": MI_PREC_BF16, "bfloat16": MI_PREC_BF16, "bf16x3": MI_PREC_BF16X3, "fp8": MI_PREC_FP8,
              "f16": MI_PREC_F16, "fp16": MI_PREC_F16, "float16": MI_PREC_F16, "f16x3": MI_PREC_F16X3}
# names that ask for "fp32 results" without insisting on exact fp32 products (see resolve_precision)
F32_NAMES = ("f32", "fp32", "float32")


def resolve_precision(precision: str, bilinear_with_weight: bool, shapes=(), concat_hidden=None) -> int:
    """Precision name -> C-ABI code.  "f32" (the default of the Python interface: the reference is fp32 throughout) means
    "results within the stated fp32 tolerances" (DESIGN.md section 2).  For the bilinear critic those tolerances are met
    by MI_PREC_BF16X3 -- every operand as two bf16 parts, three bf16 MFMAs per product, fp32 accumulation -- at 6 - 7 times
    the speed of the exact fp32-input MFMA (0.35 ms against 2.3 ms per step at B = 4096, d = 512), so that is what "f32"
    runs there when every size is a multiple of 8.  For the reference's make_mlp critic (``concat_hidden`` = (h1, h2)) they
    are met by MI_PREC_F16X3 -- two-part fp16 operands, three MFMAs per product in the forward, two in the backward
    kernels -- at a third of the exact mode's time (105 ms against >= 337 ms per step at B = 4096, h = 1024 / 512); that is
    what "f32" runs there on the shapes of the fused kernels (h1 % 64 == 0, h2 in {256, 512}).  "f32_exact" insists on exact
    fp32 products (v_mfma_f32_32x32x2_f32).  Other critics and shapes: "f32" is the exact mode."""
    if precision not in PRECISIONS:
        raise ValueError(f"unknown precision {precision!r}: expected one of {sorted(PRECISIONS)}")
    if precision in F32_NAMES and bilinear_with_weight and shapes and all(int(v) % 8 == 0 for v in shapes):
        return MI_PREC_BF16X3
    if precision in F32_NAMES and concat_hidden is not None:
        h1, h2 = (int(v) for v in concat_hidden)
        if h1 >= 64 and h1 % 64 == 0 and h2 in (256, 512):
            return MI_PREC_F16X3
    return PRECISIONS[precision]
STATS_BYTES = 64
RECORD_FLOATS = 8

# name -> (restype, argtypes); must list every symbol of include/mi_critic.h (tests/test_abi.py checks this)
_P, _I64, _I, _SZ = c_void_p, c_int64, c_int, c_size_t
SIGNATURES = {
    "mi_abi_version": (c_int, []),
    "mi_last_error": (c_char_p, []),
    "mi_profile_begin": (c_int, []),
    "mi_profile_end": (c_int, [c_char_p, _SZ, ctypes.POINTER(c_float), _I, ctypes.POINTER(c_int)]),
    "mi_bound_workspace_bytes": (_SZ, [_I64]),
    "mi_bound_fwd": (c_int, [_P, _I64, _I64, _I, _P, _P, _P, _SZ, _P]),
    "mi_bound_bwd": (c_int, [_P, _I64, _I64, _P, _P, _P, _P]),
    "mi_matrix_bound_workspace_bytes": (_SZ, [_I64]),
    "mi_matrix_bound_fwd": (c_int, [_P, _P, _I64, _I, _P, _P, _P, _SZ, _P]),
    "mi_matrix_bound_bwd": (c_int, [_P, _P, _I64, _P, _P, _P, _P]),
    "mi_pairs_count_host": (c_int, [_P, _I64, ctypes.POINTER(c_int64), _P, _SZ, _P]),
    "mi_pair_index_workspace_bytes": (_SZ, [_I64]),
    "mi_pair_index": (c_int, [_P, _I64, _P, _P, _I64, _P, _P, _P, _SZ, _P]),
    "mi_create_pairs": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P]),
    "mi_create_pairs_bwd": (c_int, [_P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "mi_bilinear_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I64, _I]),
    "mi_bilinear_fwd": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "mi_bilinear_bwd": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I, _P, _P, _P, _P, _P, _P, _SZ, _I, _P]),
    "mi_bilinear_raw_records": (_SZ, [_I64, _I64, _I64, _I64, _I, _P]),
    "mi_bilinear_bwd_records": (c_int, [_P] * 5 + [_I64] * 5 + [_I, _I, _P, _I64, _I64] + [_P] * 7 + [_SZ, _P]),
    "mi_bilinear_bwd_dw": (c_int, [_I64, _I64, _I64, _I64, _I, _P, _P, _SZ, _P]),
    "mi_bilinear_prep_local": (c_int, [_P, _P, _I64, _I64, _I64, _I64, _I, _P, _SZ, _P]),
    "mi_bilinear_fp8_stage": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I, _P, _P, _SZ, _P]),
    "mi_bilinear_step": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "mi_bilinear_step_bf16": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _SZ, _P]),
    "mi_bilinear_path": (c_int, [_I64, _I64, _I64, _I64, _I]),
    "mi_separable_path": (c_int, [_I64, _I64, _I64, _I64, _I64, _I]),
    "mi_separable_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I64, _I64, _I]),
    "mi_separable_step": (c_int, [_P] * 5 + [_I64] * 4 + [_I, _I] + [_P] * 9 + [_SZ, _P]),
    "mi_separable_fwd": (c_int, [_P] * 6 + [_I64] * 6 + [_I, _I, _I] + [_P] * 4 + [_SZ, _P]),
    "mi_separable_bwd": (c_int, [_P] * 6 + [_I64] * 6 + [_I] + [_P] * 7 + [_SZ, _I, _P]),
    "mi_concat_mlp_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I64, _I64, _I64, _I, _I]),
    "mi_concat_mlp_fwd": (c_int, [_P] * 10 + [_I64] * 7 + [_I, _I, _I] + [_P] * 5 + [_SZ, _P]),
    "mi_concat_mlp_bwd": (c_int, [_P] * 10 + [_I64] * 7 + [_I] + [_P] * 12 + [_SZ, _P]),
    "mi_merge_partials": (c_int, [_P, _I64, _I64, _I, _P, _P, _P]),
}

_lib: Optional[ctypes.CDLL] = None


MI_ESHAPE = -2  # include/mi_critic.h


class MiCriticError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library, failing loudly when it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise MiCriticError(
            f"HIP library not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {os.path.join(_PKG_ROOT, 'csrc')}`. There is no CPU fallback for the MI critic path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


MI_PATH_GENERIC, MI_PATH_GEMMS, MI_PATH_FUSED, MI_PATH_FUSED_TAIL, MI_PATH_FP8_GEMMS = 0, 1, 2, 3, 4
PATH_NAMES = {0: "generic strided-operand kernels", 1: "16-bit GEMM chain with G / G^T through HBM",
              2: "fused B x B kernel, three-launch tail", 3: "fused B x B kernel, two-launch tail", 4: "fp8 GEMM chain"}
_warned_paths = set()


def note_path(kind: str, shape: tuple, precision: int) -> int:
    """The kernel path a shape takes (MI_PATH_*), with ONE logging.warning per (critic, shape, precision) when a 16-bit
    call leaves the fused kernels: such shapes are correct but 2 - 3 times slower and move G / G^T through HBM."""
    lib = load()
    path = lib.mi_bilinear_path(*shape, precision) if kind == "bilinear" else lib.mi_separable_path(*shape, precision)
    key = (kind, tuple(shape), precision)
    if precision in (MI_PREC_BF16,) and 0 <= path < MI_PATH_FUSED and key not in _warned_paths:
        _warned_paths.add(key)
        import logging
        logging.getLogger("mutual_info_img_txt").warning(
            "%s critic, shape (b_rows, b, widths) = %s: outside the fused B x B kernel (needs batch %% 32 == 0 and a "
            "score width in {128, 256, 512, 768, 1024}); running the %s", kind, tuple(shape), PATH_NAMES.get(path, path))
    return path


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    msg = load().mi_last_error()
    msg = msg.decode() if msg else ""
    if rc in (-1, -2):
        raise ValueError(f"{what}: {msg} (code {rc})")
    raise MiCriticError(f"{what}: {msg} (code {rc})")


def stream_ptr(device=None) -> int:
    """hipStream_t of torch's current stream on ``device`` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


def call(name: str, device, *args) -> None:
    """Run the C-ABI entry point ``name(*args, stream)`` with ``device`` current and torch's current stream of THAT
    device as the trailing stream argument (the library never sets the device itself: tensors on cuda:1 while cuda:0 is
    current would otherwise be launched on the wrong device's stream).  Raises on a non-zero status."""
    lib = load()
    with torch.cuda.device(device):
        rc = getattr(lib, name)(*args, torch.cuda.current_stream(device).cuda_stream)
    check(rc, name)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_device(t: torch.Tensor, name: str) -> None:
    if not torch.is_tensor(t):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise MiCriticError(f"{name} is on {t.device}: the MI critic path runs only on a ROCm device "
                            "(no CPU fallback; the CPU oracle under oracle/ is test infrastructure)")


def f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    require_device(t, name)
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype}); the reference path is fp32 throughout")
    return t.contiguous()


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def new_stats(device) -> torch.Tensor:
    return torch.empty(STATS_BYTES, dtype=torch.uint8, device=device)


class kernel_profile:
    """Context manager: per-kernel HIP-event timings of every library launch inside the block.
    ``.records`` is a list of (name, milliseconds) in launch order; ``.by_name()`` aggregates."""

    def __init__(self, capacity: int = 65536):
        self.capacity = capacity
        self.records = []

    def __enter__(self):
        check(load().mi_profile_begin(), "mi_profile_begin")
        return self

    def __exit__(self, *exc):
        names = ctypes.create_string_buffer(self.capacity * 48)
        ms = (c_float * self.capacity)()
        n = c_int(0)
        check(load().mi_profile_end(names, len(names), ms, self.capacity, ctypes.byref(n)), "mi_profile_end")
        parts = names.raw.split(b"\0")
        self.records = [(parts[k].decode(), float(ms[k])) for k in range(n.value)]
        return False

    def by_name(self) -> dict:
        agg = {}
        for name, t in self.records:
            a = agg.setdefault(name, {"calls": 0, "ms_total": 0.0})
            a["calls"] += 1
            a["ms_total"] += t
        for a in agg.values():
            a["ms_avg"] = a["ms_total"] / a["calls"]
        return agg


def stats_dict(stats: torch.Tensor) -> dict:
    """Host copy of a statistics block (synchronises)."""
    raw = stats.cpu()
    f = raw[:32].view(torch.float32)
    i = raw[32:].view(torch.int64)
    return {"lse": float(f[0]), "pos_mean": float(f[1]), "loss_dv": float(f[2]), "loss_infonce": float(f[3]),
            "log_n_neg": float(f[4]), "neg_max": float(f[5]), "n_neg": int(i[0]), "n_pos": int(i[1])}
