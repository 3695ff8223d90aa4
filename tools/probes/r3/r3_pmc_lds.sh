#!/bin/bash
# LDS bank-conflict share of the policy step's kernels (eager launches): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_lds
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o pmc -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --no-graph ${EXTRA:-} > $OUT/a.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    key = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
print("kernel | launches | lds conflict / lds active | lds_active / gui | gui total")
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:28]:
    n = cnt[(key, "GRBM_GUI_ACTIVE")] or 1
    print(f"{key:60s} n={n:4d} conflict {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):5.3f}  lds_active/gui {c['SQ_LDS_IDX_ACTIVE']/max(c['GRBM_GUI_ACTIVE'],1):7.3f}  gui {c['GRBM_GUI_ACTIVE']/8:10.0f}")
PY
rm -rf $OUT/a
