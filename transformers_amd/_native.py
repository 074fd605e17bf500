"""Loader of libtamd_torch.so -- the compiled `torch.ops.tamd.*` (csrc/torch_binding.cpp).

`torch.ops.load_library` registers the op schemas and their CUDA (= HIP) / CPU implementations; the three plain-C entry
points next to them are bound with ctypes: `bind` points the ops at a C-ABI kernel library (libtamd.so, libtamd_diag.so,
or the CPU execution model of the test-suite), `gemm_log` / `gemm_log_summary` drive the HIP-event log around the MFMA-GEMM
launches that bench.py's `roofline` object is computed from.
"""
from __future__ import annotations

import ctypes
from pathlib import Path

import torch

from ._cabi import TamdError

LIB_PATH = Path(__file__).resolve().parent / "libtamd_torch.so"
_dll = None


def load():
    """Load the compiled ops (idempotent).  Raises with the build command when the library is missing."""
    global _dll
    if _dll is not None:
        return _dll
    from . import build

    # a fresh checkout (no library), or a library built from another source / against another torch (an ABI bump, a torch
    # upgrade: the stamp next to it says so): (re)build -- 15 s of host C++, no GPU and no hipcc needed; atomic and
    # serialised across ranks (build.build_torch_binding).  A library without a stamp cannot be checked and is loaded as is.
    current = build.torch_binding_is_current() if LIB_PATH.exists() else False
    if current is False:
        try:
            build.build_torch_binding()
        except Exception as e:
            what = "not found" if not LIB_PATH.exists() else "is stale (built from another source or torch version)"
            raise TamdError(f"{LIB_PATH} {what} and building it failed ({e}); run `python -m transformers_amd.build` "
                            "(the compiled torch.ops.tamd.* binding; the MI355X path has no Python/eager fallback)") from e
    torch.ops.load_library(str(LIB_PATH))
    dll = ctypes.CDLL(str(LIB_PATH))
    dll.tamd_torch_bind.restype = ctypes.c_int
    dll.tamd_torch_bind.argtypes = [ctypes.c_char_p, ctypes.c_int]
    dll.tamd_torch_last_error.restype = ctypes.c_char_p
    dll.tamd_torch_bound_path.restype = ctypes.c_char_p
    dll.tamd_torch_dw_cut.restype = ctypes.c_int
    dll.tamd_torch_dw_cut.argtypes = [ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong),
                                      ctypes.c_int]
    dll.tamd_torch_set_dw_balance.restype = ctypes.c_int
    dll.tamd_torch_set_dw_balance.argtypes = [ctypes.c_int]
    dll.tamd_torch_swiglu_bwd_choice.restype = ctypes.c_int
    dll.tamd_torch_swiglu_bwd_choice.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    dll.tamd_torch_gemm_log.argtypes = [ctypes.c_int]
    dll.tamd_torch_gemm_log_summary.restype = ctypes.c_int
    dll.tamd_torch_gemm_log_summary.argtypes = [ctypes.POINTER(ctypes.c_double)]
    _dll = dll
    return dll


def bind(path, emulated: bool = False) -> None:
    """Point the compiled ops at the C-ABI library at `path` (every symbol of include/tamd.h, same ABI version)."""
    dll = load()
    if dll.tamd_torch_bind(str(path).encode(), int(bool(emulated))) != 0:
        raise TamdError(dll.tamd_torch_last_error().decode())


def bound_path() -> str:
    return load().tamd_torch_bound_path().decode()


def dw_cut(m: int, n: int, k: int, cus: int = 256):
    """(axis, at) of the cut `gemm_dw_balanced` (csrc/torch_binding.cpp) makes in a weight-gradient product dW[m, n] over k
    tokens on a device of `cus` compute units: axis 0 = rows [0, at) / [at, m) as two launches, 1 = columns, -1 = one launch."""
    at = ctypes.c_longlong(0)
    axis = load().tamd_torch_dw_cut(m, n, k, ctypes.byref(at), cus)
    return axis, at.value


def set_dw_balance(on: bool) -> bool:
    """A/B switch of the weight-gradient cut (tools/gemm_dw_cut_ab.py); returns the previous setting."""
    return bool(load().tamd_torch_set_dw_balance(int(bool(on))))


def gemm_log(on: bool) -> None:
    """Start (clearing what was recorded) / stop the HIP-event log around every MFMA-GEMM launch."""
    load().tamd_torch_gemm_log(int(bool(on)))


def swiglu_bwd_choices():
    """The measured choices between the two forms of the SiLU*up backward (csrc/torch_binding.cpp swiglu_bwd_fused), one dict per
    shape seen: tokens, intermediate, hidden, form, and the two timings (0.0: pinned by TAMD_FUSE_SWIGLU_BWD or unmeasured)."""
    dll, out, recs = load(), (ctypes.c_double * 6)(), []
    for i in range(dll.tamd_torch_swiglu_bwd_choice(-1, None)):
        dll.tamd_torch_swiglu_bwd_choice(i, out)
        recs.append(dict(tokens=int(out[0]), intermediate=int(out[1]), hidden=int(out[2]),
                         form="dX GEMM way out" if out[3] else "GEMM + swiglu_bwd kernel", fused_ms=out[4], two_kernels_ms=out[5]))
    return recs


def gemm_log_summary():
    """-> dict(launches, flops, ms, bytes; fused_bwd_launches / _flops / _ms: the launches among them that carry the SiLU*up
    backward in their way out) of the recorded launches (synchronises their events, clears the log)."""
    out = (ctypes.c_double * 8)()
    if load().tamd_torch_gemm_log_summary(out) != 0:
        raise TamdError("reading the GEMM event log failed")
    return dict(launches=int(out[0]), flops=out[1], ms=out[2], bytes=out[3], fused_bwd_launches=int(out[4]),
                fused_bwd_flops=out[5], fused_bwd_ms=out[6])
