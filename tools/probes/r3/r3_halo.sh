#!/bin/bash
# halo_probe under the env settings given in CFGS ("name:ENV=V,ENV=V" ...) + optional LDS-conflict counters
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in ${CFGS:-base:}; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "== $name ($envs)"
  env $(echo $envs | tr ',' ' ') timeout 300 python tools/probes/halo_probe.py 2>&1 | grep -v amdgpu.ids
done
if [ -n "$PMC" ]; then
  export TMPDIR=/tmp; cd /tmp
  OUT=$R/gpurun_out/pmc_halo; rm -rf $OUT; mkdir -p $OUT
  env $(echo $PMC_ENV | tr ',' ' ') timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/a -o pmc -- python $R/tools/probes/halo_probe.py pmc > $OUT/a.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    if "conv_" not in k: continue
    key = (k, r.get("Grid_Size", ""))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
for key, c in sorted(acc.items()):
    n = cnt[(key, "GRBM_GUI_ACTIVE")] or 1
    print(f"{key[0]:60s} grid {key[1]:>8s} n={n:3d} ldsconf/active {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):5.3f}  lds_active/gui/256CU {c['SQ_LDS_IDX_ACTIVE']/n/max(c['GRBM_GUI_ACTIVE']/n,1)/256:5.3f} mfma_util {c['SQ_VALU_MFMA_BUSY_CYCLES']/n/max(c['GRBM_GUI_ACTIVE']/n*1024,1):5.3f} wait_lds/wave_cycles {c['SQ_WAIT_INST_LDS']/max(c['SQ_WAVE_CYCLES'],1):5.3f}")
PY
  rm -rf $OUT/a
fi
