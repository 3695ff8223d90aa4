"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass.
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the chip's SIMDs (MI355X_MICROARCH.md: = 32 x N_mfma for 32x32x16
bf16); GRBM_GUI_ACTIVE comes back accumulated over the 8 XCDs (each has its own GRBM), so the launch's clock cycles are
GRBM_GUI_ACTIVE / 8 and utilisation = busy / (active / 8 * 1024 SIMDs) -- cross-checked against achieved / peak TFLOP/s from HIP events
(fp32 video conv: 0.81 here vs 122 / 157.3 = 0.78).  Usage: mfma_util.py <dir> -> JSON."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

SIMDS = 256 * 4
XCDS = 8
root = sys.argv[1]
acc = defaultdict(lambda: {"busy": 0.0, "active": 0.0, "n": 0})
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"] or 0)
    for (_, name), c in per.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        short = name.split("(")[0].replace("void ", "")
        a = acc[short]
        a["busy"] += c["SQ_VALU_MFMA_BUSY_CYCLES"]
        a["active"] += c["GRBM_GUI_ACTIVE"]
        a["n"] += 1
out = {}
for k, a in sorted(acc.items(), key=lambda kv: -kv[1]["active"]):
    if a["active"] <= 0 or a["busy"] <= 0:
        continue
    out[k] = {"launches": a["n"], "mfma_util": a["busy"] / (a["active"] / XCDS * SIMDS), "share_of_active_cycles": a["active"]}
tot = sum(v["share_of_active_cycles"] for v in out.values()) or 1.0
for v in out.values():
    v["share_of_active_cycles"] /= tot
print(json.dumps(dict(list(out.items())[:16]), indent=1))
