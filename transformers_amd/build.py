"""Build the native libraries in-tree: libtamd.so (the gfx950 C-ABI kernel library, hipcc) and libtamd_torch.so (the
compiled `torch.ops.tamd.*` binding over it, host C++ against the torch headers).

`python -m transformers_amd.build [--force]` or `transformers_amd.build.build()`.
hipcc cross-compiles for gfx950 without a GPU; the resulting `.so` files sit next to this
file (git-ignored, but they travel with the tree to the GPU box).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
INCLUDE = HERE.parent / "include"
LIB = HERE / "libtamd.so"
DIAG_LIB = HERE / "libtamd_diag.so"  # diagnostics only (include/tamd_diag.h): tools/ and tests/test_gpu_probe.py
OBJ_DIR = HERE / "_build"
SOURCES = ["api.hip", "norm.hip", "elementwise.hip", "gemm.hip", "gemv.hip", "attention.hip", "attention_bwd_dkdv.hip", "optim.hip"]
DIAG_SOURCES = SOURCES + ["probe.hip"]  # + every source recompiled with -DTAMD_DIAG (ablation instantiations, traces)
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wno-unused-result", "-I", str(CSRC), "-I", str(INCLUDE)] + os.environ.get("TAMD_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


# per-source extra flags, overridable for experiments: TAMD_FLAGS_<stem>="..."
# attention.hip: keep MFMA results in arch VGPRs (the softmax reads every S/P element with VALU ops; the default
# AGPR form costs a v_accvgpr_read/write per element): +39 % forward, +11 % backward on MI355X.  The same option
# crashes clang 22 (ROCm 7.2) on gemm.hip, where it would not matter (accumulators are only touched by MFMA).
PER_SOURCE_FLAGS: dict = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
                          "attention_bwd_dkdv.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"]}


def _compile(hipcc: str, src: Path, obj: Path, verbose: bool, defines=()) -> None:
    extra = os.environ.get(f"TAMD_FLAGS_{src.stem}", None)
    extra = extra.split() if extra is not None else PER_SOURCE_FLAGS.get(src.name, [])
    cmd = [hipcc, *FLAGS, *defines, *extra, "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")


def _build(lib: Path, sources, obj_dir: Path, defines, force: bool, verbose: bool) -> Path:
    srcs = [CSRC / s for s in sources if (CSRC / s).exists()]
    # everything a translation unit can include: headers AND the .inc kernel bodies (attention_bwd.inc)
    deps = srcs + sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc")) + sorted(INCLUDE.glob("*.h"))
    stamp = obj_dir / "stamp"
    digest = _digest(deps) + " ".join(defines)
    if not force and lib.exists() and stamp.exists() and stamp.read_text() == digest:
        return lib
    hipcc = _hipcc()
    obj_dir.mkdir(parents=True, exist_ok=True)
    objs = [obj_dir / (s.stem + ".o") for s in srcs]
    # per object: recompiled when ITS source, any header / .inc (conservatively: all of them), the flags or the defines changed --
    # gemm.hip alone takes ~13 minutes, and a change to norm.hip should not pay for it
    shared = [d for d in deps if d not in srcs]

    def obj_digest(src: Path) -> str:
        extra = os.environ.get(f"TAMD_FLAGS_{src.stem}", None)
        extra = extra.split() if extra is not None else PER_SOURCE_FLAGS.get(src.name, [])
        return _digest([src] + shared) + " ".join(defines) + " ".join(extra)

    def one(so) -> None:
        src, obj = so
        ostamp = obj.with_suffix(".stamp")
        d = obj_digest(src)
        if not force and obj.exists() and ostamp.exists() and ostamp.read_text() == d:
            return
        _compile(hipcc, src, obj, verbose, defines)
        ostamp.write_text(d)

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        list(ex.map(one, zip(srcs, objs)))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(lib), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(digest)
    return lib


TORCH_LIB = HERE / "libtamd_torch.so"  # torch.ops.tamd.*: csrc/torch_binding.cpp (host C++ only, no device code)


def torch_binding_source_digest() -> str:
    """What a built libtamd_torch.so must match to be loaded: its source, the C ABI header and the torch it was compiled
    against (cheap: no compiler probing -- `_native.load()` checks it at import)."""
    import torch

    return _digest([CSRC / "torch_binding.cpp", INCLUDE / "tamd.h"]) + " torch " + torch.__version__


def torch_binding_is_current() -> bool | None:
    """True / False: the library on disk was built from the current source and torch; None: no stamp to tell by."""
    stamp = OBJ_DIR / "torch_binding.stamp"
    if not TORCH_LIB.exists() or not stamp.exists():
        return None
    lines = stamp.read_text().split("\n")
    return len(lines) >= 1 and lines[0] == torch_binding_source_digest()


def build_torch_binding(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/torch_binding.cpp against the installed torch (TORCH_LIBRARY(tamd, ...): the compiled dispatcher ops
    that call the C ABI of include/tamd.h through a table bound at run time).  Safe under concurrent callers (several
    ranks importing a fresh checkout): one builds under a file lock, into a temporary file that replaces the library
    atomically -- nobody can `load_library` a half-written .so."""
    import fcntl

    import torch
    from torch.utils import cpp_extension as ce

    src = CSRC / "torch_binding.cpp"
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no C++ compiler found for the torch binding")
    torch_lib = Path(torch.__file__).resolve().parent / "lib"
    rocm = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-unused-result"]
    for inc in ce.include_paths():
        cmd += ["-isystem", inc]
    cmd += ["-isystem", str(rocm / "include"), str(src)]
    tail = [f"-L{torch_lib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", f"-L{rocm / 'lib'}",
            "-lamdhip64", "-ldl", f"-Wl,-rpath,{torch_lib}", f"-Wl,-rpath,{rocm / 'lib'}"]
    stamp = OBJ_DIR / "torch_binding.stamp"
    digest = torch_binding_source_digest() + "\n" + " ".join(cmd + tail)
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    with open(OBJ_DIR / "torch_binding.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)  # (released when the file is closed)
        if not force and TORCH_LIB.exists() and stamp.exists() and stamp.read_text() == digest:
            return TORCH_LIB  # up to date -- possibly built by the rank that held the lock before us
        tmp = TORCH_LIB.with_name(f".{TORCH_LIB.name}.{os.getpid()}.tmp")
        full = cmd + ["-o", str(tmp)] + tail
        if verbose:
            print(" ".join(full), flush=True)
        r = subprocess.run(full, capture_output=True, text=True)
        if r.returncode != 0:
            tmp.unlink(missing_ok=True)
            raise RuntimeError(f"building {TORCH_LIB.name} failed:\n{r.stdout}\n{r.stderr[-6000:]}")
        os.replace(tmp, TORCH_LIB)
        stamp.write_text(digest)
    return TORCH_LIB


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 and link libtamd.so (the product library), then the compiled torch binding
    over it (libtamd_torch.so).  Returns the kernel library path."""
    lib = _build(LIB, SOURCES, OBJ_DIR, (), force, verbose)
    build_torch_binding(force, verbose)
    return lib


def build_diag(force: bool = False, verbose: bool = False) -> Path:
    """libtamd_diag.so: the same sources with -DTAMD_DIAG (ablation instantiations, phase traces) + probe.hip."""
    return _build(DIAG_LIB, DIAG_SOURCES, OBJ_DIR / "diag", ("-DTAMD_DIAG",), force, verbose)


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
    if "--diag" in sys.argv:
        print(build_diag(force="--force" in sys.argv, verbose=True))
