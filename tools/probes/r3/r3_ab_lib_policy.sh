#!/bin/bash
# same-box A/B of two builds of the library on the policy step (fp32 and bf16): libv2a_hip_alt.so (baseline) vs libv2a_hip.so
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
L=video-to-action-release_amd/v2a_hip
cp $L/libv2a_hip.so $L/libv2a_hip_new.so
for r in $(seq 1 ${REPS:-2}); do
  for which in alt new; do
    cp $L/libv2a_hip_$which.so $L/libv2a_hip.so
    for P in fp32 bf16; do
      python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --precision $P 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which $P', round(d['ms_per_step'],3), 'ms')"
    done
  done
done
cp $L/libv2a_hip_new.so $L/libv2a_hip.so
