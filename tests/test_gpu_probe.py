"""Silicon vs the CPU execution model: the gfx950 instruction semantics that tests/hipemu implements
(MFMA fragment layouts, ds_read_b64_tr_b16, v_permlane32_swap, direct-to-LDS loads, ballot) are run on
the GPU through tamd_probe() and compared bit for bit with the model's answer for the same operands."""
import ctypes

import numpy as np
import pytest
import torch

from transformers_amd import _cabi, build


def _run(lib, which, inp, in2, dev, dtype_code=0):
    tin = torch.from_numpy(inp.view(np.int32)).to(dev)
    tin2 = torch.from_numpy(in2.view(np.int32)).to(dev)
    out = torch.zeros(4096, dtype=torch.int32, device=dev)
    stream = None if dev == "cpu" else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.tamd_probe(ctypes.c_void_p(tin.data_ptr()), ctypes.c_void_p(tin2.data_ptr()),
                        ctypes.c_void_p(out.data_ptr()), which, dtype_code, stream)
    assert rc == 0
    if dev != "cpu":
        torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)


def _inputs(which, rng):
    inp = rng.integers(0, 2 ** 32, size=4096, dtype=np.uint64).astype(np.uint32)
    in2 = np.zeros(64, dtype=np.uint32)
    if which in (0, 1):
        # small-integer bf16 operands: products and sums are exact in fp32 whatever the accumulation order
        vals = rng.integers(-4, 5, size=8192).astype(np.float32)
        bits = (vals.view(np.uint32) >> 16).astype(np.uint16)
        inp = bits.view(np.uint32).copy()
    elif which == 2:
        # per-lane 8-byte-aligned byte offsets: the strided/swizzled pattern of frag_tr plus random ones
        in2 = (rng.integers(0, 2048, size=64).astype(np.uint32)) * 8
    elif which in (4, 5):
        in2 = (rng.permutation(512)[:64].astype(np.uint32)) * 16
    elif which == 6:
        in2 = (rng.permutation(500)[:64].astype(np.uint32)) * 16  # + immediates up to 3072 + 2048: inside 16 KiB
    elif which == 7:
        in2 = (rng.permutation(512)[:64].astype(np.uint32)) * 16  # up to 8 KiB: half the lanes past the 4 KiB / 2 KiB buffers
    return inp, in2


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 2, 3, 4, 5, 6, 7])
def test_hardware_matches_cpu_model(which):
    from emu_backend import get_emu

    hip = _cabi.TamdLib(build.build_diag(), diag=True)  # diagnostics library (include/tamd_diag.h)
    emu = get_emu().lib
    rng = np.random.default_rng(which)
    for trial in range(4):
        inp, in2 = _inputs(which, rng)
        got = _run(hip, which, inp, in2, "cuda:0")
        want = _run(emu, which, inp, in2, "cpu")
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (f"probe {which} trial {trial}: {bad.size} words differ, first at {bad[:8]}: "
                               f"hw {got[bad[:8]]} model {want[bad[:8]]}")


@pytest.mark.gpu
def test_probe_f16_mfma():
    from emu_backend import get_emu

    hip = _cabi.TamdLib(build.build_diag(), diag=True)  # diagnostics library (include/tamd_diag.h)
    emu = get_emu().lib
    rng = np.random.default_rng(99)
    vals = rng.integers(-4, 5, size=8192).astype(np.float16)
    inp = vals.view(np.uint16).view(np.uint32).copy()
    in2 = np.zeros(64, dtype=np.uint32)
    for which in (0, 1):
        assert np.array_equal(_run(hip, which, inp, in2, "cuda:0", 1), _run(emu, which, inp, in2, "cpu", 1))
