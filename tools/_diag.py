"""Route transformers_amd.ops through libtamd_diag.so (include/tamd_diag.h): the kernel sources built with
-DTAMD_DIAG (ablation instantiations selected by TAMD_GEMM_DBG / TAMD_DKDV_DBG, phase traces) plus the probes.
Tools only -- the product package never loads this library."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import _cabi, build, ops  # noqa: E402


class DiagBackend(ops.HipBackend):
    def __init__(self):
        self.lib = _cabi.TamdLib(build.build_diag(), diag=True)


def use_diag():
    ops._set_backend(DiagBackend())
    return ops.backend().lib
