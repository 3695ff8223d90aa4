#!/bin/bash
# MFMA-utilisation / stall counters of the contraction kernels on the micro-benchmark shapes (run via gpurun).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_conv
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o pmc -- python $R/tools/conv_bench.py > $OUT/a.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:52], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
print("kernel | grid | MfmaUtil% (MFMA_BUSY/(GUI_ACTIVE*1024 SIMDs)) | wait_any/wave_cycles | wait_inst/wave_cycles | active/wave_cycles | lds_conflict/lds_active")
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:24]:
    n = cnt[(key, "GRBM_GUI_ACTIVE")] or 1
    gui = c.get("GRBM_GUI_ACTIVE", 0) / n
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n
    wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{key[0]:52s} {key[1]:>9s} util {100*mf/max(gui*1024,1):6.1f}%  wait_any {c.get('SQ_WAIT_ANY',0)/wc:5.2f}  wait_inst {c.get('SQ_WAIT_INST_ANY',0)/wc:5.2f}  active {c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f}  ldsconf {c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',0),1):5.3f}  gui_active {gui:10.0f}")
PY
rm -rf $OUT/a
