"""Test configuration.

`-m "not gpu"` : CPU suite -- oracle vs golden vectors, host logic, C-ABI surface, and the kernel sources
                 executed under the lane-accurate CPU model in tests/hipemu (same .hip files, host clang).
`-m gpu`       : parity tests proper -- the same test bodies through libtamd.so on an MI355X.
"""
import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hipemu"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs the HIP kernels through the C-ABI)")


class Env:
    """What a kernel test needs: the device to allocate on and the backend context."""

    def __init__(self, name):
        self.name = name
        self.device = torch.device("cuda:0") if name == "hip" else torch.device("cpu")
        # the CPU model is ~1e5x slower than the GPU: tests scale their shapes with this flag
        self.big = name == "hip"


def _backend_params():
    return [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=_backend_params())
def env(request):
    from transformers_amd import ops

    name = request.param
    if name == "hip":
        if not torch.cuda.is_available():
            pytest.fail("GPU test selected but no GPU is visible")
        old = ops._set_backend(None)  # force the real HipBackend (raises if libtamd.so is missing)
        try:
            ops.backend()
            yield Env("hip")
        finally:
            ops._set_backend(old)
    else:
        from emu_backend import emu_backend

        with emu_backend():
            yield Env("emu")


@pytest.fixture(scope="session", autouse=True)
def _build_library_once():
    """Make sure libtamd.so exists (hipcc cross-compiles without a GPU); the emulator builds lazily."""
    from transformers_amd import build

    if not build.LIB.exists():
        build.build()
    yield


class Measured(float):
    """An error that remembers what it was compared with: `assert rel_err(a, b) < 4e-3` records (measured, limit, where)
    for the parity report (profiles/r02_parity.json: every GPU test's measured error next to its gate)."""

    def _note(self, limit):
        import inspect

        fr = inspect.currentframe().f_back.f_back
        key = f"{Path(fr.f_code.co_filename).name}:{fr.f_lineno} ({fr.f_code.co_name})"
        e = _GATES.setdefault(key, {"err": 0.0, "limit": float(limit), "n": 0})
        e["err"] = max(e["err"], float(self))
        e["limit"] = max(e["limit"], float(limit))
        e["n"] += 1

    def __lt__(self, limit):
        self._note(limit)
        return float(self) < limit

    def __le__(self, limit):
        self._note(limit)
        return float(self) <= limit


_GATES = {}


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return Measured(((a - b).norm() / b.norm().clamp_min(1e-30)).item())


def max_err(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


# ---- measured errors ("dump measured rel-errors of every gpu test to profiles/r02_parity.json") -------------
_PARITY = {}


def record(group, name, err, ref=None):
    """Remember a measured error (and, for noise-floor gates, the reference's own bf16 error) for the parity report."""
    _PARITY.setdefault(group, {})[name] = {"err": float(err)} if ref is None else {"err": float(err), "ref": float(ref)}


def pytest_sessionfinish(session, exitstatus):
    if _GATES:
        _PARITY["gates"] = _GATES
    if not _PARITY:
        return
    import json

    backend = "hip" if torch.cuda.is_available() else "emu"
    out = ROOT / "gpurun_out" if backend == "hip" else ROOT / ".pytest_cache"
    out.mkdir(exist_ok=True)
    path = out / f"parity_{backend}.json"
    merged = {}
    if path.exists():  # a partial session (-k ...) must not drop what a full one measured
        try:
            merged = json.loads(path.read_text())
        except ValueError:
            merged = {}
    for grp, entries in _PARITY.items():
        merged.setdefault(grp, {}).update(entries)  # the latest measurement (and gate) of an entry wins
    with open(path, "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)
