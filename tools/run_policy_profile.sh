#!/bin/bash
# usage: run_policy_profile.sh [fp32|bf16] [tag]   -- on the GPU box: rocprofv3 kernel trace of the policy bench + phase timeline
PREC=${1:-fp32}; TAG=${2:-p}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o policy -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --precision $PREC > $R/gpurun_out/$TAG.log 2>&1
cd $R; f=$(find gpurun_out/$TAG -name "*kernel_trace.csv" | head -1)
python tools/phase_timeline.py $f 6; python tools/timeline.py $f 6 | head -40
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
