"""Is the GEMM bounded by the power budget?  The gate|up forward product (32768 x 28672 x 4096) in a steady loop of ~4 s per arm --
the three-barrier K loop (product), the one-barrier ring (diagnostic library, tamd_gemm_set_dbg(2048)), torch.mm (hipBLASLt), and the
product kernel on ZERO operands (no bit toggles in the multipliers) -- while a sampler thread reads the board's power, clocks and
temperature through rocm-smi every 0.25 s; the kernel's own clock probe (s_memtime / s_memrealtime around the K loop of every
workgroup) gives the clock and the pipe-busy share of the last launch of each arm.

    python tools/gemm_power.py > gpurun_out/<tag>_gemm_power.jsonl
"""
import ctypes
import json
import re
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import _diag  # noqa: E402
from transformers_amd import ops  # noqa: E402

lib = _diag.use_diag()
dev = torch.device("cuda:0")
M, N, K = 32768, 28672, 4096
SECONDS = 4.0


def smi_sample():
    """One reading: watts, sclk MHz, temperature (whatever this rocm-smi prints; missing fields stay None)."""
    out = {"W": None, "sclk_MHz": None, "temp_C": None}
    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True,
                           text=True, timeout=5)
        d = json.loads(r.stdout)
        card = next(iter(d.values()))
        for k, v in card.items():
            kl = k.lower()
            if "power" in kl and out["W"] is None:
                m = re.search(r"[\d.]+", str(v))
                out["W"] = float(m.group()) if m else None
            if kl.startswith("sclk") and out["sclk_MHz"] is None:
                m = re.search(r"\((\d+)Mhz\)", str(v))
                out["sclk_MHz"] = float(m.group(1)) if m else None
            if "temperature" in kl and "junction" in kl:
                m = re.search(r"[\d.]+", str(v))
                out["temp_C"] = float(m.group()) if m else None
    except Exception as e:  # the tool reports what it can
        out["error"] = repr(e)
    return out


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.samples = []

    def run(self):
        while not self.stop:
            s = smi_sample()
            s["t"] = time.perf_counter()
            self.samples.append(s)
            time.sleep(0.25)


def mean(xs):
    xs = [x for x in xs if x is not None]
    return round(sum(xs) / len(xs), 1) if xs else None


def arm(name, fn, probe=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    smp = Sampler()
    smp.start()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    smp.stop = True
    smp.join()
    ms = e0.elapsed_time(e1) / n
    # the first second is the ramp (clock and power settle): report the rest
    late = [s for s in smp.samples if s["t"] - t0 > 1.0]
    rec = {"arm": name, "launches": n, "ms": round(ms, 4), "TF": round(2.0 * M * N * K / ms / 1e9),
           "W": mean([s["W"] for s in late]), "sclk_MHz": mean([s["sclk_MHz"] for s in late]),
           "temp_C": mean([s["temp_C"] for s in late]), "samples": len(late)}
    if probe is not None:
        rec.update(probe())
    print(json.dumps(rec), flush=True)


def clock_probe(a, b):
    wgs = (M // 256) * (N // 256)
    buf = torch.zeros(2 * wgs, dtype=torch.int64, device=dev)
    lib.tamd_gemm_set_clock_buffer(ctypes.c_void_p(buf.data_ptr()))
    ops.raw_gemm(a, b)
    torch.cuda.synchronize()
    lib.tamd_gemm_set_clock_buffer(ctypes.c_void_p(0))
    t = buf.cpu().view(wgs, 2).double()
    return {"clock_GHz": round((t[:, 0] / t[:, 1]).mean().item() * 0.1, 4),
            "mfma_busy_in_k_loop": round(((K // 64) * 2048 / t[:, 0]).mean().item(), 4)}


torch.manual_seed(0)
x = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
xz, wz = torch.zeros_like(x), torch.zeros_like(w)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
print(json.dumps({"idle": smi_sample()}), flush=True)
for rnd in range(2):
    lib.tamd_gemm_set_dbg(0)
    arm("three-barrier loop (product)", lambda: ops.raw_gemm(x, w), lambda: clock_probe(x, w))
    lib.tamd_gemm_set_dbg(2048)
    arm("one-barrier ring (round 5)", lambda: ops.raw_gemm(x, w), lambda: clock_probe(x, w))
    lib.tamd_gemm_set_dbg(0)
    arm("torch.mm (hipBLASLt)", lambda: torch.mm(x, w.t(), out=out))
    arm("product kernel, zero operands", lambda: ops.raw_gemm(xz, wz), lambda: clock_probe(xz, wz))
