#!/bin/bash
# A/B of an env switch on the policy step (fp32 + bf16): r3_ab.sh VAR val1 val2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
VAR=$1; shift
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-video --no-predict --no-roofline-pass --no-bf16-extra"
for rep in 1 2; do
for v in "$@"; do
  for prec in ${PRECS:-fp32 bf16}; do
  env $VAR=$v $B --precision $prec 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v $prec %.3f ms (median %.3f)  loss %.5f' % (d['ms_per_step'], d['ms_per_step_median_hip_events'], d['final_loss']))
"
  done
done
done
