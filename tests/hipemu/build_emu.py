"""Build libtamd_emu.so: the SAME kernel sources as libtamd.so, compiled by host clang against the
CPU execution model in tests/hipemu (TEST INFRASTRUCTURE -- never loaded by the product package)."""
from __future__ import annotations

import hashlib
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "transformers_amd" / "csrc"
OUT = HERE / "_build"
LIB = OUT / "libtamd_emu.so"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SOURCES = ["api.hip", "norm.hip", "elementwise.hip", "gemm.hip", "gemv.hip", "attention.hip", "attention_bwd_dkdv.hip", "optim.hip", "probe.hip"]
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-DTAMD_DIAG", "-ffp-contract=off", "-pthread", "-Wno-unused-value",
         "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-pass-failed",
         "-I", str(HERE), "-I", str(CSRC), "-I", str(ROOT / "include")]


def build(force: bool = False) -> Path:
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()] + [HERE / "emu_api.cpp"]
    deps = srcs + sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc")) + sorted(HERE.glob("*.h")) + [HERE / "hip" / "hip_runtime.h"]
    h = hashlib.sha256()
    for p in deps:
        h.update(p.name.encode() + p.read_bytes())
    digest = h.hexdigest()
    stamp = OUT / "stamp"
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == digest:
        return LIB
    OUT.mkdir(exist_ok=True)
    objs = [OUT / (s.stem + ".o") for s in srcs]

    def cc(so):
        r = subprocess.run([CLANG, *FLAGS, "-c", str(so[0]), "-o", str(so[1])], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"emu compile failed for {so[0].name}:\n{r.stderr[-6000:]}")

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(cc, zip(srcs, objs)))
    r = subprocess.run([CLANG, "-shared", "-pthread", "-o", str(LIB), *map(str, objs)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"emu link failed:\n{r.stderr}")
    stamp.write_text(digest)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
