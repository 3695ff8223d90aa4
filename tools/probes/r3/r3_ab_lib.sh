#!/bin/bash
# same-box A/B of two builds of the library on the sampler leg: libv2a_hip_alt.so (baseline) vs libv2a_hip.so; alternating, REPS rounds
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
L=video-to-action-release_amd/v2a_hip
cp $L/libv2a_hip.so $L/libv2a_hip_new.so
for r in $(seq 1 ${REPS:-2}); do
  for which in alt new; do
    cp $L/libv2a_hip_$which.so $L/libv2a_hip.so
    timeout 600 python tools/video_only.py --storage ${STORAGE:-bf16} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', round(d['value'],2), 'frames/s', d['seconds_per_sample_call'])"
  done
done
cp $L/libv2a_hip_new.so $L/libv2a_hip.so
