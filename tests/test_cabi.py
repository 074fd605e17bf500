"""The C-ABI surface: every symbol declared in include/tamd.h is exported by libtamd.so and bound with
the declared arity; argument errors are reported without launching anything (no GPU needed)."""
import ctypes
import re
from pathlib import Path

import pytest

from transformers_amd import _cabi, build

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "tamd.h").read_text()


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|uint32_t|const char\*)\s+(tamd_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_header_matches_binding_table():
    decl = declared_functions()
    assert set(decl) == set(_cabi.SIGNATURES), set(decl) ^ set(_cabi.SIGNATURES)
    for name, n in decl.items():
        assert len(_cabi.SIGNATURES[name][1]) == n, name


def test_library_exports_every_symbol():
    lib_path = build.build()
    dll = ctypes.CDLL(str(lib_path))
    for name in declared_functions():
        assert hasattr(dll, name), f"{name} not exported by {lib_path}"
    lib = _cabi.TamdLib(lib_path)
    assert lib.tamd_abi_version() == _cabi.ABI_VERSION
    assert lib.tamd_error_string(0) == b"ok"


def test_diagnostics_are_not_in_the_product_library():
    """ablation instantiations (wrong results by design), phase traces and probes are built only into
    libtamd_diag.so (include/tamd_diag.h, -DTAMD_DIAG); libtamd.so neither exports nor contains them."""
    dll = ctypes.CDLL(str(build.build()))
    for name in _cabi.DIAG_SIGNATURES:
        assert not hasattr(dll, name), name
        assert name not in HEADER
    diag_header = (ROOT / "include" / "tamd_diag.h").read_text()
    diag = _cabi.TamdLib(build.build_diag(), diag=True)
    for name in _cabi.DIAG_SIGNATURES:
        assert name in diag_header and hasattr(diag, name)
    blob = Path(build.build()).read_bytes()
    assert b"TAMD_GEMM_DBG" not in blob and b"TAMD_DKDV_DBG" not in blob
    assert b"TAMD_GEMM_DBG" in Path(build.build_diag()).read_bytes()


def test_build_digest_covers_included_kernel_bodies(tmp_path, monkeypatch):
    """attention_bwd.inc is part of libtamd.so; an edit there must invalidate the build stamp."""
    srcs = [build.CSRC / s for s in build.SOURCES]
    deps = srcs + sorted(build.CSRC.glob("*.h")) + sorted(build.CSRC.glob("*.inc")) + sorted(build.INCLUDE.glob("*.h"))
    assert any(p.name == "attention_bwd.inc" for p in deps)
    d0 = build._digest(deps)
    inc = tmp_path / "attention_bwd.inc"
    inc.write_bytes((build.CSRC / "attention_bwd.inc").read_bytes() + b"\n// edit\n")
    d1 = build._digest([p if p.name != "attention_bwd.inc" else inc for p in deps])
    assert d0 != d1


def test_argument_errors_do_not_launch():
    lib = _cabi.TamdLib(build.build())
    # NULL pointers / bad shapes are rejected before any launch (safe without a GPU)
    assert lib.tamd_rmsnorm_fwd(None, None, None, None, None, None, 4, 64, 1e-5, 0, None) == -4
    assert lib.tamd_gemm(None, None, None, None, None, 8, 8, 8, 8, 8, 8, 0, 0, 0, 0, 0, None) == -4
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.cast(buf, ctypes.c_void_p)
    aligned = ctypes.c_void_p((p.value + 15) & ~15)
    assert lib.tamd_gemm(aligned, aligned, aligned, None, None, 8, 8, 7, 8, 8, 8, 0, 0, 0, 0, 0, None) == -2  # K % 8
    assert lib.tamd_gemm(aligned, aligned, aligned, None, None, 8, 8, 8, 8, 8, 8, 0, 0, 0, 0, 2, None) == -1  # fp32
    ap = _cabi.AttnParams()
    ap.q = ap.k = ap.v = ap.o = aligned.value
    ap.batch, ap.heads_q, ap.heads_kv, ap.seq_q, ap.seq_k, ap.head_dim = 1, 2, 1, 8, 8, 80
    assert lib.tamd_attn_fwd(ctypes.byref(ap), None) == -2  # head_dim 80 unsupported
    # row strides: the tile loaders address 64 rows with 32-bit offsets from a scalar base and cut ragged tiles off with the
    # buffer's size -- rows must follow each other upwards, at most 2^24 elements apart (a one-row operand's stride is ignored)
    ap.head_dim = 64
    for name, bad in (("k_stride_s", 0), ("v_stride_s", -64), ("q_stride_s", 1 << 25), ("o_stride_s", 32)):
        for f in ("q_stride_s", "k_stride_s", "v_stride_s", "o_stride_s"):
            setattr(ap, f, 128)
        for f in ("q_stride_b", "k_stride_b", "v_stride_b", "o_stride_b", "q_stride_h", "k_stride_h", "v_stride_h", "o_stride_h"):
            setattr(ap, f, 64)
        ap.dtype = _cabi.TAMD_BF16
        setattr(ap, name, bad)
        assert lib.tamd_attn_fwd(ctypes.byref(ap), None) == -6, name


def test_product_path_has_no_cpu_fallback():
    import torch

    from transformers_amd import ops

    old = ops._set_backend(None)
    try:
        # the compiled op refuses a CPU tensor while the product library is bound (csrc/torch_binding.cpp `Launch`)
        with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
            ops.raw_rmsnorm_fwd(torch.randn(4, 64).bfloat16(), torch.ones(64).bfloat16(), 1e-5)
    finally:
        ops._set_backend(old)
