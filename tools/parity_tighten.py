"""Measured errors of every test gate (tests/conftest.py `Measured`: each `rel_err(...) < limit` comparison records
itself) from the GPU run (gpurun_out/parity_hip.json) and the CPU-model run (.pytest_cache/parity_emu.json):
  * writes profiles/<round>_parity.json (--round r03, default r02) -- measured error, gate, ratio, for both backends;
    --hip FILE takes the GPU measurements from a saved copy (profiles/r03c_parity.json) instead of gpurun_out/;
  * with --apply, rewrites a gate's literal in the test source when it is more than 2x the measured error
    (new gate = 2 x measured, rounded up to two significant digits; never below 1e-5; never loosens)."""
import json
import math
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
def _arg(flag, default):
    return sys.argv[sys.argv.index(flag) + 1] if flag in sys.argv else default


srcs = {"hip": Path(_arg("--hip", str(ROOT / "gpurun_out" / "parity_hip.json"))),
        "emu": ROOT / ".pytest_cache" / "parity_emu.json"}
data = {k: json.loads(p.read_text()) if p.exists() else {} for k, p in srcs.items()}
gates = {}
for be, d in data.items():
    for key, g in d.get("gates", {}).items():
        e = gates.setdefault(key, {"limit": g["limit"]})
        e[be] = g["err"]
        e["limit"] = min(e["limit"], g["limit"])  # (gates only ever got tighter: the smaller one is the current one)


def round_up(x):
    if x <= 0:
        return 0.0
    mag = 10 ** (math.floor(math.log10(x)) - 1)
    return math.ceil(x / mag) * mag


report, edits = {}, []
for key, g in sorted(gates.items()):
    m = max(g.get("hip", 0.0), g.get("emu", 0.0))
    new = max(round_up(2.0 * m), 1e-5)
    report[key] = {"measured_hip": g.get("hip"), "measured_emu": g.get("emu"), "gate": g["limit"],
                   "gate_over_measured": (g["limit"] / m) if m > 0 else None}
    if m > 0 and g["limit"] > 2.0 * m * 1.0001 and new < g["limit"]:
        edits.append((key, g["limit"], new))
out = {"gates": report}
for be, d in data.items():
    for grp, v in d.items():
        if grp != "gates":
            out.setdefault("noise_floor_gates_" + be, {})[grp] = v
(ROOT / "profiles" / f"{_arg('--round', 'r02')}_parity.json").write_text(json.dumps(out, indent=1, sort_keys=True))
print(f"{len(report)} gates, {len(edits)} looser than 2x the measured error")
if "--apply" in sys.argv:
    done = 0
    for key, old, new in edits:
        fname, rest = key.split(":", 1)
        line = int(rest.split(" ")[0])
        path = ROOT / "tests" / fname
        lines = path.read_text().split("\n")
        txt = lines[line - 1]
        lits = [m for m in re.finditer(r"(?<![\w.])(\d+\.?\d*e-?\d+|\d*\.\d+)(?![\w.])", txt) if abs(float(m.group(1)) - old) < 1e-12 * max(1, old)]
        if len(lits) != 1:
            print("  skip (literal not unique on the line):", key, old, "->", new)
            continue
        m = lits[0]
        lines[line - 1] = txt[:m.start(1)] + f"{new:.2g}" + txt[m.end(1):]
        path.write_text("\n".join(lines))
        done += 1
    print(f"rewrote {done} gates")
else:
    for key, old, new in edits:
        print(f"  {key}: gate {old:g} -> {new:.2g}")
