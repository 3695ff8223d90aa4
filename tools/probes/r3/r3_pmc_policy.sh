#!/bin/bash
# stall / MFMA-utilisation counters of the policy step's kernels (eager launches so that every dispatch is its own record)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_pol
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o pmc -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --no-graph ${EXTRA:-} > $OUT/a.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Grid_Size", r.get("Grid_Size_X", "")))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
print("kernel | grid | launches | MfmaUtil | wait_any/wave | wait_inst/wave | wait_lds/wave | active/wave | gui_active(avg)")
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:30]:
    n = cnt[(key, "GRBM_GUI_ACTIVE")] or 1
    gui = c.get("GRBM_GUI_ACTIVE", 0) / n
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n
    wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{key[0]:48s} {key[1]:>8s} n={n:4d} util {100*mf/max(gui/8*1024,1):5.1f}%  wait_any {c.get('SQ_WAIT_ANY',0)/wc:5.2f}  wait_inst {c.get('SQ_WAIT_INST_ANY',0)/wc:5.2f}  lds {c.get('SQ_WAIT_INST_LDS',0)/wc:5.2f}  active {c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f}  gui {gui/8:9.0f}")
PY
rm -rf $OUT/a
