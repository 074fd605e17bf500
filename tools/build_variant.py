"""Build a VARIANT of libtamd.so for a side-by-side A/B through the C ABI (tools/attn_variants_ab.py):
    python tools/build_variant.py <name> [-DMACRO=1 ...] [--flags-<stem>="..."] [--base-<stem>]
compiles csrc/attention.hip and csrc/attention_bwd_dkdv.hip with the extra defines (experiment switches `TAMD_X_*`, which
exist in the sources only while an experiment runs; `--base-attention` / `--base-attention_bwd_dkdv` take that object from
tools/ab/_obj_head/ instead -- a copy of the objects of the commit the experiment started from -- so that one kernel changes at a
time) and links them with the product build's other objects into
tools/ab/libtamd_<name>.so (git-ignored; it travels with the tree to the GPU box).  CPU only (hipcc cross-compiles gfx950)."""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from transformers_amd import build as tb  # noqa: E402

VARIANT_SOURCES = ["attention.hip", "attention_bwd_dkdv.hip"]


def main():
    name = sys.argv[1]
    defines = [a for a in sys.argv[2:] if a.startswith("-D")]
    per = {a.split("=", 1)[0][len("--flags-"):]: a.split("=", 1)[1].split() for a in sys.argv[2:] if a.startswith("--flags-")}
    tb.build()  # the product objects the variant links against
    out_dir = ROOT / "tools" / "ab" / f"_obj_{name}"
    out_dir.mkdir(parents=True, exist_ok=True)
    hipcc = tb._hipcc()

    base = {a[len("--base-"):] for a in sys.argv[2:] if a.startswith("--base-")}

    def comp(src):
        if Path(src).stem in base:
            return ROOT / "tools" / "ab" / "_obj_head" / (Path(src).stem + ".o")
        extra = per.get(Path(src).stem, tb.PER_SOURCE_FLAGS.get(src, []))
        obj = out_dir / (Path(src).stem + ".o")
        cmd = [hipcc, *tb.FLAGS, *defines, *extra, "-c", str(tb.CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{src}: {r.stderr}")
        return obj

    with ThreadPoolExecutor(2) as ex:
        objs = list(ex.map(comp, VARIANT_SOURCES))
    others = [tb.OBJ_DIR / (Path(s).stem + ".o") for s in tb.SOURCES if s not in VARIANT_SOURCES]
    lib = ROOT / "tools" / "ab" / f"libtamd_{name}.so"
    r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={tb.ARCH}", "-o", str(lib), *map(str, objs + others)],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    print(lib, lib.stat().st_size)


if __name__ == "__main__":
    main()
