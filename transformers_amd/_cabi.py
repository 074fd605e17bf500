"""ctypes binding of libtamd.so -- the C-ABI declared in include/tamd.h.

This file is the *only* place that knows the symbol signatures on the Python side; it is a
literal transcription of include/tamd.h.  tests/test_cabi.py checks that every symbol the header
declares is exported by the library and listed here.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint8, c_void_p
from pathlib import Path

TAMD_BF16, TAMD_F16, TAMD_F32 = 0, 1, 2
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_QUICK_GELU, ACT_SILU = 0, 1, 2, 3, 4
GEMM_A_KM, GEMM_B_KN = 1, 2
EPI_NONE, EPI_BIAS, EPI_RESIDUAL, EPI_BIAS_ACT, EPI_ACCUM = 0, 1, 2, 3, 4
ABI_VERSION = 11

P = c_void_p
I64 = c_int64


class AttnParams(Structure):
    _fields_ = [
        ("q", P), ("k", P), ("v", P), ("o", P), ("lse", P), ("key_valid", P),
        ("batch", I64), ("heads_q", I64), ("heads_kv", I64), ("seq_q", I64), ("seq_k", I64), ("head_dim", I64),
        ("q_stride_b", I64), ("q_stride_s", I64), ("q_stride_h", I64),
        ("k_stride_b", I64), ("k_stride_s", I64), ("k_stride_h", I64),
        ("v_stride_b", I64), ("v_stride_s", I64), ("v_stride_h", I64),
        ("o_stride_b", I64), ("o_stride_s", I64), ("o_stride_h", I64),
        ("scale", c_float), ("causal", c_int), ("dtype", c_int),
        ("dropout_p", c_float), ("dropout_seed", ctypes.c_uint64), ("q_start", P), ("q_prescaled", ctypes.c_int32),
        ("dropout_seed_dev", P),
    ]


class GemmProblem(Structure):
    """tamd_gemm_problem (include/tamd.h): one product of a grouped launch."""
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("c", c_void_p), ("m", c_int64), ("n", c_int64), ("k", c_int64),
                ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64)]


class AttnBwdParams(Structure):
    _fields_ = [("fwd", AttnParams), ("dout", P), ("dq", P), ("dk", P), ("dv", P), ("delta", P), ("rope_cos", P),
                ("rope_sin", P), ("rope_cos_batch", I64)]


# name -> (restype, argtypes); order and types mirror include/tamd.h
SIGNATURES = {
    "tamd_abi_version": (c_int, []),
    "tamd_error_string": (c_char_p, [c_int]),
    "tamd_rmsnorm_fwd": (c_int, [P, P, P, P, P, P, I64, I64, c_float, c_int, P]),
    "tamd_norm_bwd_workspace_bytes": (c_size_t, [I64, I64]),
    "tamd_rmsnorm_bwd": (c_int, [P, P, P, P, P, P, P, P, c_size_t, I64, I64, c_int, P]),
    "tamd_layernorm_fwd": (c_int, [P, P, P, P, P, P, P, P, I64, I64, c_float, c_int, P]),
    "tamd_layernorm_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_size_t, I64, I64, c_int, P]),
    "tamd_layernorm_dropout_fwd": (c_int, [P, P, P, P, P, P, P, P, I64, I64, c_float, c_float, ctypes.c_uint64, P, c_int, P]),
    "tamd_layernorm_dropout_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, c_size_t, I64, I64, c_float,
                                           ctypes.c_uint64, P, c_int, P]),
    "tamd_rope_inplace": (c_int, [P, P, P, I64, I64, I64, I64, I64, I64, c_int, I64, c_float, c_int, P]),
    "tamd_embedding_fwd": (c_int, [P, P, P, I64, I64, I64, P, c_int, P]),
    "tamd_embedding_bwd_workspace_bytes": (c_size_t, [I64, I64]),
    "tamd_embedding_bwd": (c_int, [P, P, P, P, P, c_size_t, I64, I64, I64, I64, c_int, P]),
    "tamd_bert_embeddings_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, c_float,
                                         c_int, P]),
    "tamd_swiglu_fwd": (c_int, [P, P, P, I64, I64, I64, I64, c_int, P]),
    "tamd_swiglu_bwd": (c_int, [P, P, P, P, P, P, I64, I64, I64, I64, c_int, P]),
    "tamd_bias_act_fwd": (c_int, [P, P, P, I64, I64, c_int, c_int, P]),
    "tamd_bias_act_bwd": (c_int, [P, P, P, P, P, P, c_size_t, I64, I64, c_int, c_int, P]),
    "tamd_add": (c_int, [P, P, P, I64, c_int, P]),
    "tamd_adamw_step": (c_int, [P, P, P, P, I64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                ctypes.c_double, I64, ctypes.c_double, c_int, c_int, P]),
    "tamd_mt_sumsq": (c_int, [P, c_int, I64, P, c_int, P]),
    "tamd_mt_norm_finish": (c_int, [P, I64, P, ctypes.c_double, P]),
    "tamd_mt_scale": (c_int, [P, c_int, I64, P, c_int, P]),
    "tamd_mt_adamw_step": (c_int, [P, c_int, I64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, I64, ctypes.c_double, P, c_int, c_int, P]),
    "tamd_colsum_workspace_bytes": (c_size_t, [I64, I64]),
    "tamd_colsum": (c_int, [P, P, P, c_size_t, I64, I64, I64, c_int, P]),
    "tamd_transpose": (c_int, [P, P, I64, I64, I64, I64, c_int, P]),
    "tamd_cross_entropy_fwd": (c_int, [P, P, P, P, I64, I64, I64, I64, c_int, P]),
    "tamd_cross_entropy_bwd": (c_int, [P, P, P, P, P, I64, I64, I64, I64, c_int, P]),
    "tamd_gemm": (c_int, [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, c_int, c_int, c_int, c_int, P]),
    "tamd_gemm_workspace_bytes": (c_size_t, [I64, I64, I64, c_int, c_int]),
    "tamd_gemm_ws": (c_int, [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, c_int, c_int, c_int, c_int, P, c_size_t,
                             P]),
    "tamd_gemm_seg": (c_int, [P, P, P, P, c_int, I64, I64, I64, I64, I64, c_int, c_int, P, c_size_t, P]),
    "tamd_gemm_group_workspace_bytes": (c_size_t, [P, c_int, c_int]),
    "tamd_gemm_group": (c_int, [P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "tamd_gemm_colscale": (c_int, [P, P, P, P, I64, I64, I64, I64, I64, I64, c_int, I64, c_float, c_int, P]),
    "tamd_gemm_swiglu": (c_int, [P, P, P, P, I64, I64, I64, I64, I64, I64, I64, c_int, P]),
    "tamd_gemm_swiglu_bwd": (c_int, [P, P, P, P, I64, I64, I64, I64, I64, I64, I64, c_int, P]),
    "tamd_dropout_hash": (ctypes.c_uint32, [ctypes.c_uint64, ctypes.c_uint64]),
    "tamd_attn_dropout_field": (ctypes.c_uint32, [ctypes.c_uint64] * 6),
    "tamd_attn_fwd": (c_int, [POINTER(AttnParams), P]),
    "tamd_attn_decode_workspace_bytes": (c_size_t, [POINTER(AttnParams)]),
    "tamd_attn_decode": (c_int, [POINTER(AttnParams), P, c_size_t, P]),
    "tamd_attn_bwd": (c_int, [POINTER(AttnBwdParams), P]),
}

# include/tamd_diag.h -- exported by libtamd_diag.so only (tools/, tests/test_gpu_probe.py), never by the product library
DIAG_SIGNATURES = {
    "tamd_gemm_trace": (c_int, [P, P, P, I64, I64, I64, P, P]),
    "tamd_gemm_set_clock_buffer": (c_int, [P]),
    "tamd_gemm_set_timeline_buffer": (c_int, [P]),
    "tamd_gemm_set_dbg": (c_int, [c_int]),
    "tamd_attn_set_trace": (c_int, [P]),
    "tamd_mfma_power": (c_int, [P, c_int, c_int, c_int, P, P, P]),
    "tamd_probe": (c_int, [P, P, P, c_int, c_int, P]),
    "tamd_bw_probe": (c_int, [P, c_size_t, c_int, c_size_t, c_int, c_int, c_int, P, P]),
}


class TamdError(RuntimeError):
    pass


class TamdLib:
    """A loaded C-ABI library with typed entry points (`lib.tamd_gemm(...)`)."""

    def __init__(self, path, diag: bool = False, accept_abi=None):
        self.path = str(path)
        self._dll = ctypes.CDLL(self.path)
        missing = []
        table = dict(SIGNATURES, **DIAG_SIGNATURES) if diag else SIGNATURES
        for name, (res, args) in table.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        if missing and not accept_abi:  # (a build of an earlier ABI lacks the newer entry points)
            raise TamdError(f"{self.path} does not export: {', '.join(missing)}")
        ver = self.tamd_abi_version()
        if ver != ABI_VERSION and ver not in (accept_abi or ()):  # (tools/attn_lib_ab.py loads a build of an earlier ABI)
            raise TamdError(f"{self.path}: ABI version {ver}, expected {ABI_VERSION}")

    def check(self, code: int, what: str) -> None:
        if code != 0:
            msg = self.tamd_error_string(code).decode()
            raise TamdError(f"{what} failed: {msg} (code {code})")


def default_library_path() -> Path:
    return Path(__file__).resolve().parent / "libtamd.so"
