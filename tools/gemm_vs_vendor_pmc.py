"""Counter-level side-by-side of `tamd::gemm_fl_kernel` and hipBLASLt's kernel (what `torch.mm` picks) on the forward
products of the Llama-3-8B layer.

    workload   : python tools/gemm_vs_vendor_pmc.py run            (a few launches of each arm per shape; run it under
                 `rocprofv3 --pmc <set> --kernel-trace --output-format csv -d <dir>/<pass> -o p`, one pass per counter set)
    the passes : python tools/gemm_vs_vendor_pmc.py passes         (prints one counter set per line -- gfx950 slots: SQ 8, TCC 4
                 (FETCH_SIZE = 3, WRITE_SIZE = 2), GRBM 2; MI355X_MICROARCH.md "rocprofv3 PMC slots")
    the table  : python tools/gemm_vs_vendor_pmc.py table <dir> [out.md]

Both arms run under the same profiler in the same process, interleaved per shape (a profiled arm is never compared with an
un-profiled one: profiled passes serialise the dispatches and run at another clock).  Effective clock = GRBM_GUI_ACTIVE /
dispatch duration (GRBM_GUI_ACTIVE is collected in every pass so each pass carries its own clock).  SQ_* cycle counters are
quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles."""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
T = 32768
#         label      M  N      K      (forward products x . W^T of the layer; gate|up is the shape where the two are level)
SHAPES = [("down", T, 4096, 14336), ("o_proj", T, 4096, 4096), ("gate_up", T, 28672, 4096), ("qkv", T, 6144, 4096)]
PASSES = [
    "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY "
    "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum",
    "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM "
    "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum",
    "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD "
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES FETCH_SIZE",
    "GRBM_GUI_ACTIVE WRITE_SIZE TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TA_BUSY_avr",
    "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum",
]


def run():
    import torch

    sys.path.insert(0, str(ROOT))
    from transformers_amd import ops

    dev = torch.device("cuda:0")
    only = set(sys.argv[2:])
    for label, m, n, k in SHAPES:
        if only and label not in only:
            continue
        torch.manual_seed(0)
        x = torch.randn(m, k, device=dev).bfloat16()
        w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
        wt = w.t()
        print(json.dumps({"shape": label, "m": m, "n": n, "k": k, "grid_256": (m // 256) * (n // 256)}), flush=True)
        for _ in range(3):  # interleaved: ours, vendor, ours, vendor, ...
            ops.raw_gemm(x, w)
            torch.mm(x, wt)
        torch.cuda.synchronize()
        del x, w, wt


def _arm(name):
    return "ours" if "tamd::gemm" in name else ("vendor" if ("Cijk" in name or "MT256" in name) else None)


def table(root, out=None):
    """One section per shape (keyed on the order the dispatches were issued: 3 x (ours, vendor) per shape, see run()); per counter
    the mean per dispatch of each arm and vendor / ours."""
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))  # shape -> arm -> counter
    for pdir in sorted(glob.glob(f"{root}/pass*")):
        disp = {}  # dispatch id -> (arm, duration ns)
        for f in glob.glob(f"{pdir}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                a = _arm(r["Kernel_Name"])
                if a:
                    disp[r["Dispatch_Id"]] = (a, float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        ctr = collections.defaultdict(dict)  # dispatch id -> counter -> value (summed over the rows of a dispatch)
        arm_of = {}
        for f in glob.glob(f"{pdir}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                a = _arm(r["Kernel_Name"])
                if a:
                    arm_of[r["Dispatch_Id"]] = a
                    c = ctr[r["Dispatch_Id"]]
                    c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        order = sorted(arm_of, key=int)  # issue order: shape s = dispatches 6s .. 6s+5
        for i, did in enumerate(order):
            shape = SHAPES[min(i // 6, len(SHAPES) - 1)][0]
            for n, v in ctr[did].items():
                per[shape][arm_of[did]][n].append(v)
            if did in disp:
                per[shape][arm_of[did]]["dispatch_us"].append(disp[did][1] / 1e3)
                if "GRBM_GUI_ACTIVE" in ctr[did]:
                    per[shape][arm_of[did]]["GRBM_GUI_ACTIVE per us"].append(ctr[did]["GRBM_GUI_ACTIVE"] / (disp[did][1] / 1e3))
    lines = ["# `tamd::gemm_fl_kernel` vs hipBLASLt's kernel: rocprofv3 PMC, same process, interleaved dispatches", ""]
    for shape, m, n, k in SHAPES:
        if shape not in per:
            continue
        lines += [f"## {shape}: {m} x {n} x {k}", "", "| counter (mean per dispatch) | ours | vendor | vendor / ours |", "|---|---|---|---|"]
        names = sorted(set(per[shape]["ours"]) | set(per[shape]["vendor"]))
        for c in names:
            o, v = per[shape]["ours"].get(c), per[shape]["vendor"].get(c)
            om = sum(o) / len(o) if o else None
            vm = sum(v) / len(v) if v else None
            lines.append(f"| {c} | {om:.6g} | {vm:.6g} | {vm / om:.3f} |" if om and vm else f"| {c} | {om} | {vm} | |")
        lines.append("")
    text = "\n".join(lines) + "\n"
    if out:
        Path(out).write_text(text)
    print(text)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "run"
    if mode == "run":
        run()
    elif mode == "passes":
        print("\n".join(PASSES))
    else:
        table(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
