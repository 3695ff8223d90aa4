#!/bin/bash
# usage: r3_trace.sh <tag> [fp32|bf16]  (env selects the variant) -- kernel trace of the policy step + flat timeline of one step
TAG=${1:-t}; PREC=${2:-fp32}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o policy -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --precision $PREC > $R/gpurun_out/$TAG.log 2>&1
cd $R; f=$(find gpurun_out/$TAG -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py $f 2 > gpurun_out/$TAG.step.txt
python tools/phase_timeline.py $f 6 > gpurun_out/$TAG.phase.txt
python tools/timeline.py $f 6 > gpurun_out/$TAG.tl.txt
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
head -3 gpurun_out/$TAG.tl.txt
