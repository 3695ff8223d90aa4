#!/bin/bash
# A/B of the two-workgroups-per-CU halo instance (V2A_H3_PAIR): parity tests under PAIR=2, then the bf16 sampler leg for each setting
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V2A_H3_PAIR=2 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "halo_h3" 2>&1 | tail -5 > gpurun_out/pair_tests.txt
for P in ${PAIRS:-0 1 2}; do
  V2A_H3_PAIR=$P timeout 600 python tools/video_only.py --storage bf16 2>gpurun_out/pair_$P.err | tail -1 > gpurun_out/pair_$P.json
  python - <<PY
import json
d=json.load(open("gpurun_out/pair_$P.json"))
print("PAIR=$P", d["value"], "frames/s", d["seconds_per_sample_call"])
for k,v in d.get("roofline",{}).get("all_conv_variants",{}).items():
    if "halo" in k: print("   ",k,v)
PY
done
cat gpurun_out/pair_tests.txt
