"""Static check of the generated gfx950 code for the one hazard the CPU execution model cannot see.

The dK/dV attention kernel and the k-major GEMM paths issue their LDS reads through inline asm ("untracked", see
csrc/tamd_device.h) and wait for them with explicit `s_waitcnt lgkmcnt(N)`.  The compiler believes the destination
registers are valid as soon as the asm statement is over; if register pressure makes it copy or spill one of them
(v_accvgpr_write, scratch_store, any use) BEFORE the covering wait, it captures stale data -- silently, and only on
the GPU.  (Seen once: a variant with three more live registers spilled a fragment to scratch and produced wrong dK.)
This test compiles the sources to assembly and checks, kernel by kernel and basic block by basic block, that no
instruction (spills and AGPR copies included) reads the destination of an untracked LDS read before a wait that
covers it."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "transformers_amd" / "csrc"

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def _self_loops_twice(lines):
    """A basic block that branches back to its own label (the hand-placed tile loops: one block per loop) is laid out twice in a
    row, so that reads requested at its bottom for the top of the next trip (the fragments of the next tile's first MFMA groups)
    meet the waits that cover them: the straight-line analysis below then sees the loop-carried reads too."""
    out, i = [], 0
    while i < len(lines):
        t = lines[i].strip()
        if t.startswith(".LBB") and ":" in t:
            label = t.split(":")[0]
            j = i + 1
            while j < len(lines) and not (lines[j].strip().startswith(".LBB") and ":" in lines[j]):
                j += 1
            block = lines[i:j]
            back = [k for k, b in enumerate(block) if re.match(r"\s*s_c?branch\w*\s+" + re.escape(label) + r"\b", b)]
            if back:
                # first trip (label, body up to and including the branch), then the body again WITHOUT the label (state carried)
                out += block[:back[-1]] + block[1:]
            else:
                out += block
            i = j
        else:
            out.append(lines[i])
            i += 1
    return out


def lint_kernel(name, lines):
    """lines: the assembly of one kernel.  Returns a list of violations."""
    pending = []  # [(set of dst regs)] of untracked LDS reads in issue order (they complete in order)
    bad = []
    in_asm = False
    lines = _self_loops_twice(lines)
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith(".LBB"):
            pending = []  # straight-line analysis only: block layout in the text is not execution order (a read issued
            continue      # at the bottom of the loop is waited for at its top), so nothing is carried across labels
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if in_asm and op.startswith("ds_read"):
            dst = t.split(None, 1)[1].split(",")[0]
            pending.append(_regs(dst))
            continue
        if op == "s_waitcnt" and "lgkmcnt" in t:
            n = int(re.search(r"lgkmcnt\((\d+)\)", t).group(1))
            # LDS operations retire in order: at most n of the newest stay outstanding (tracked reads issued in between
            # only make the real wait stricter)
            pending = pending[len(pending) - n:] if n < len(pending) else pending
            if n == 0:
                pending = []
            continue
        if op in ("s_barrier", "s_endpgm") or op.startswith("s_cbranch") or op == "s_branch":
            if op != "s_barrier":
                pending = []
            continue
        if not pending:
            continue
        args = t.split(None, 1)[1] if " " in t else ""
        if op.startswith("ds_read") or op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("scratch_load"):
            srcs = _regs(args.split(",", 1)[1]) if "," in args else set()  # first operand is the destination
        else:
            srcs = _regs(args)
            if not (op.startswith("ds_write") or op.startswith("global_store") or op.startswith("scratch_store")
                    or op.startswith("buffer_store") or op.startswith("v_cmp")):
                srcs = _regs(args.split(",", 1)[1]) if "," in args else set()  # drop the destination operand
        for p in pending:
            if srcs & p:
                bad.append(f"{name}: `{t}` reads v{sorted(srcs & p)} before the wait covering its untracked read")
    return bad


@pytest.mark.timeout(900)
@pytest.mark.parametrize("src,flags,defines", [
    ("attention_bwd_dkdv.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"], []),
    ("attention.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], []),
    ("attention.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], ["-DTAMD_DIAG"]),  # + the phase-trace instrumentation
    ("gemm.hip", [], [])])
def test_untracked_lds_reads_are_not_touched_before_their_wait(src, flags, defines, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    from transformers_amd import build as tb

    assert flags == tb.PER_SOURCE_FLAGS.get(src, []), "keep this test's flags in step with build.py"
    out = tmp_path / "k.s"
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", *flags, *defines, "-I", str(CSRC), "-I",
           str(ROOT / "include"), "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", str(CSRC / src),
           "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kernels = re.split(r"\n(?=_ZN4tamd\w+:[ \t])", text)
    checked, bad = 0, []
    for k in kernels:
        m = re.match(r"(_ZN4tamd\w+):", k)
        if not m or ";;#ASMSTART\n\tds_read" not in k:
            continue
        dbg = re.search(r"ELi(\d+)ELb[01]E+vNS_11AttnBwdArgs", m.group(1))
        if dbg and dbg.group(1) != "0":
            continue  # ablation builds (TAMD_DKDV_DBG) drop reads / waits on purpose
        checked += 1
        body = k.split("s_endpgm")[0]
        bad += lint_kernel(m.group(1), body.splitlines())  # (a spill of a pending register is a read of it: caught)
    assert checked > 0
    assert not bad, "\n".join(bad[:10])
    if src == "gemm.hip":
        # a second thing only the generated code shows: per-lane software divisions in a GEMM kernel.  The tile decode
        # divides on the scalar unit; a `%` or `/` on a per-lane 64-bit index inside a way-out row loop costs ~40 VALU
        # instructions per row segment with nothing to overlap them (the first rotary epilogue did: 96 v_mul_hi_u32, 12.5 k
        # instructions, 150 us per launch slower than GEMM + rotary kernel -- profiles/r02_gemm_variants.md section 7)
        heavy = []
        for k in kernels:
            m = re.match(r"(_ZN4tamd\w*gemm_\w+):", k)
            if m:
                n = k.split("s_endpgm")[0].count("v_mul_hi_u32")
                if n > 8:
                    heavy.append(f"{m.group(1)}: {n} x v_mul_hi_u32")
        assert not heavy, "per-lane divisions in GEMM kernels:\n" + "\n".join(heavy[:10])
        # and a third: scratch.  The 256 x 256 kernel runs 256 + 256 registers per lane; an instantiation that spills is
        # throttled in how many of its waves a CU runs (and a spilled untracked fragment is the hazard above).  Round 4 shipped
        # one for a day: a second copy of the epilogue for segmented outputs cost the Llama dW kernel 4 spilled VGPRs.
        # Checked: the instantiations of the Llama step (plain / split-K / SwiGLU forward and backward (105) / rotary epilogues in every layout, the
        # row-major residual one) and the grouped launches.  (The k-major residual / accumulate epilogues hold a half's
        # residual rows beside 256 accumulators and spill 5 .. 17 registers on the way out: known, outside the K loop.)
        usage = re.findall(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", r.stderr, flags=re.S)
        assert usage

        def must_be_clean(n):
            m = re.search(r"gemm_fl_kernelINS_\w+?_tELb([01])ELb([01])ELi(\d+)ELi\d+ELi0EEEvNS_8GemmArgs", n)
            if m:
                return int(m.group(3)) in (0, 100, 101, 103, 105) or (m.group(1) == "0" and m.group(2) == "0")
            m = re.search(r"gemm_fl_group_kernelINS_\w+?_tELb[01]ELb[01]ELi(\d+)EEEvNS_13GemmGroupArgs", n)
            return bool(m) and int(m.group(1)) in (0, 100)

        seen = [n for n, _ in usage if must_be_clean(n)]
        assert len(seen) >= 8, seen
        spilled = [f"{n}: {b} B/lane" for n, b in usage if must_be_clean(n) and int(b) > 0]
        assert any("ELi105E" in n for n in seen), "the SiLU*up backward way out (tamd_gemm_swiglu_bwd) was not checked"
        assert not spilled, "full-line GEMM kernels with scratch:\n" + "\n".join(spilled)
