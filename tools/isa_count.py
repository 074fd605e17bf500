"""Instruction-class histogram of the MFMA-carrying basic blocks of one kernel in a gfx950 assembly listing -- where the
"instructions per tile" figures of DESIGN.md section 3.3 come from (attention loop bodies: 167 -> 136, 200 -> 158, 489 -> 415
with the srcC chains; the dropout variants 1206 -> 806 with one hash per 2 x 2 block).

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I transformers_amd/csrc -I include \\
        -mllvm -amdgpu-mfma-vgpr-form=1 [-fno-slp-vectorize] -S --cuda-device-only \\
        -Rpass-analysis=kernel-resource-usage transformers_amd/csrc/attention_bwd_dkdv.hip -o /tmp/dkdv.s 2> /tmp/dkdv.rpass
  python tools/isa_count.py /tmp/dkdv.s attn_bwd_dkdv_kernelINS_6bf16_tELi128ELb1ELb0ELb0ELi0ELb0 [-v] [--min-mfma 8]

(the second argument is a substring of the kernel's mangled name; -v lists the VALU opcodes of each block; registers,
spills and occupancy are in the .rpass file)."""
import collections
import sys


def cls(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith(("v_exp", "v_log", "v_rcp")):
        return "trans"
    if op.startswith("v_cvt_pk"):
        return "cvt"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


def main():
    path, sub = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[sys.argv.index("--min-mfma") + 1]) if "--min-mfma" in sys.argv else 8
    lines = open(path).read().split("\n")
    start = next((i for i, l in enumerate(lines) if l.startswith("_ZN") and sub in l.split(":")[0] and ": " in l), None)
    if start is None:
        sys.exit(f"no kernel matching {sub!r} in {path}")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blk, blocks = "entry", collections.OrderedDict()
    for l in lines[start:end]:
        t = l.strip()
        if t.startswith(".LBB"):
            blk = t.split(":")[0]
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        blocks.setdefault(blk, []).append(t.split()[0])
    for b, ops in blocks.items():
        h = collections.Counter(cls(o) for o in ops)
        if h["mfma"] >= min_mfma:
            print(b, len(ops), dict(h))
            if "-v" in sys.argv:
                print("   ", collections.Counter(o for o in ops if cls(o) == "valu").most_common(25))


if __name__ == "__main__":
    main()
