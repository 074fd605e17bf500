"""Counter-level picture of the three attention kernels at the Llama-3-8B shape (forward, dQ, dK/dV).

    the passes : python tools/attn_pmc.py passes     (one counter set per line; run the workload once per set under
                 `rocprofv3 --pmc <set> --kernel-trace --output-format csv -d <dir>/pass<i> -o p -- <workload>`, where the
                 workload is `AB_LIBS=new AB_SHAPES=llama3-8b python tools/attn_lib_ab.py`)
    the table  : python tools/attn_pmc.py table <dir> [out.md]

Per kernel the mean per dispatch of every counter, the dispatch duration and the effective clock (GRBM_GUI_ACTIVE / duration; every
pass carries GRBM_GUI_ACTIVE), and the derived shares that say where a wave's time goes: SQ_* cycle counters are quad-cycles summed
over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import sys
from pathlib import Path

PASSES = [
    "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY "
    "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA",
    "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS "
    "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU",
    "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT "
    "SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC",
]
KERNELS = {"attn_fwd_kernel": "forward", "attn_bwd_dq_kernel": "dQ", "attn_bwd_dkdv_kernel": "dK/dV"}


def _which(name):
    for k, v in KERNELS.items():
        if k in name:
            return v
    return None


def table(root, out=None):
    per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> values
    for pdir in sorted(glob.glob(f"{root}/pass*")):
        dur = {}
        for f in glob.glob(f"{pdir}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if _which(r["Kernel_Name"]):
                    dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        ctr = collections.defaultdict(dict)
        kern = {}
        for f in glob.glob(f"{pdir}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = _which(r["Kernel_Name"])
                if k:
                    kern[r["Dispatch_Id"]] = k
                    c = ctr[r["Dispatch_Id"]]
                    c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for did, k in kern.items():
            for n, v in ctr[did].items():
                per[k][n].append(v)
            if did in dur:
                per[k]["dispatch_us"].append(dur[did] / 1e3)
                if "GRBM_GUI_ACTIVE" in ctr[did]:
                    per[k]["clock_GHz"].append(ctr[did]["GRBM_GUI_ACTIVE"] / dur[did])
    lines = ["# The attention kernels at the Llama-3-8B shape (8 x 4096, 32 / 8 heads, head_dim 128, causal): rocprofv3 PMC", ""]
    mean = {k: {n: sum(v) / len(v) for n, v in d.items()} for k, d in per.items()}
    names = sorted({n for d in mean.values() for n in d})
    ks = [k for k in ("forward", "dQ", "dK/dV") if k in mean]
    lines += ["| counter (mean per dispatch) | " + " | ".join(ks) + " |", "|---|" + "---|" * len(ks)]
    for n in names:
        lines.append(f"| {n} | " + " | ".join(f"{mean[k].get(n, float('nan')):.6g}" for k in ks) + " |")
    lines += ["", "| derived | " + " | ".join(ks) + " |", "|---|" + "---|" * len(ks)]

    def row(label, fn):
        vals = []
        for k in ks:
            try:
                vals.append(f"{fn(mean[k]):.3f}")
            except (KeyError, ZeroDivisionError):
                vals.append("")
        lines.append(f"| {label} | " + " | ".join(vals) + " |")

    row("matrix pipe busy per wave cycle (MFMA_BUSY / 4 WAVE_CYCLES)", lambda m: m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_WAVE_CYCLES"]))
    row("matrix pipe busy per GPU cycle and SIMD (MFMA_BUSY / (1024 GUI_ACTIVE))", lambda m: m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"]))
    row("wave parked at s_waitcnt / s_barrier (WAIT_ANY / WAVE_CYCLES)", lambda m: m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"])
    row("wave waiting to issue (WAIT_INST_ANY / WAVE_CYCLES)", lambda m: m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"])
    row("... on LDS (WAIT_INST_LDS / WAVE_CYCLES)", lambda m: m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"])
    row("issuing (ACTIVE_INST_ANY / WAVE_CYCLES)", lambda m: m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"])
    row("instructions per MFMA (SALU + VALU + LDS + VMEM + SMEM) / MFMA", lambda m: (m["SQ_INSTS_SALU"] + m["SQ_INSTS_VALU"] + m["SQ_INSTS_LDS"] + m["SQ_INSTS_VMEM_RD"] + m["SQ_INSTS_SMEM"]) / m["SQ_INSTS_MFMA"] - 1.0)
    row("LDS bank-conflict cycles / LDS active cycles", lambda m: m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"])
    row("MFMA and VALU co-executing / MFMA busy", lambda m: m["SQ_VALU_MFMA_COEXEC_CYCLES"] / m["SQ_VALU_MFMA_BUSY_CYCLES"])
    text = "\n".join(lines) + "\n"
    if out:
        Path(out).write_text(text)
    print(text)


if __name__ == "__main__":
    if sys.argv[1] == "passes":
        print("\n".join(PASSES))
    else:
        table(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
