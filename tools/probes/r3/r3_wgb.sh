#!/bin/bash
# round 3: grouped weight-gradient launches, A/B over the grouping modes (run via gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-video --no-predict --no-roofline-pass"
CFGS=${CFGS:-0:0 stage:0 enc:0}
for cfg in $CFGS; do
  for prec in fp32 bf16; do
  V2A_WGRAD_BATCH=${cfg%%:*} V2A_WGRAD_BATCH_STREAM=${cfg##*:} $B --precision $prec 2>gpurun_out/r3_wgb.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch=$cfg $prec %.3f ms  loss %.5f' % (d['ms_per_step'], d['final_loss']))
" >> gpurun_out/r3_wgb.log 2>&1
  done
done
cat gpurun_out/r3_wgb.log
