"""CPU execution-model backend for the ops layer (TEST INFRASTRUCTURE).

`with emu_backend():` routes transformers_amd.ops through libtamd_emu.so -- the same kernel sources
compiled against tests/hipemu -- so the host logic (argument marshalling, autograd formulas, module
wiring) is exercised on CPU tensors.  Never imported by the product package.
"""
from __future__ import annotations

import contextlib
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(HERE) not in sys.path:
    sys.path.insert(0, str(HERE))

import build_emu  # noqa: E402
from transformers_amd import _cabi, ops  # noqa: E402


class EmuBackend:
    name = "emu"

    def __init__(self):
        self.lib = _cabi.TamdLib(build_emu.build(), diag=True)

    def check_tensor(self, t):
        if t.device.type != "cpu":
            raise RuntimeError("the CPU execution model needs CPU tensors")

    def stream(self, t):
        return None


_EMU = None


def get_emu() -> EmuBackend:
    global _EMU
    if _EMU is None:
        _EMU = EmuBackend()
    return _EMU


@contextlib.contextmanager
def emu_backend():
    old = ops._set_backend(get_emu())
    try:
        yield get_emu()
    finally:
        ops._set_backend(old)
