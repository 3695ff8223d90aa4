#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM-byte PMC passes for the policy step and the video sampler.
# Usage: tools/profile_round.sh r01
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/policy -o policy -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass > $OUT/policy_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/video -o video -- python $R/tools/video_only.py --steps 3 > $OUT/video_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/video_bf16 -o video -- python $R/tools/video_only.py --steps 3 --storage bf16 > $OUT/video_bf16_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/policy_bf16 -o policy -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --precision bf16 > $OUT/policy_bf16_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_policy -o pmc -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --no-graph > $OUT/pmc_fetch_policy.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_policy -o pmc -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --no-graph > $OUT/pmc_write_policy.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_policy_bf16 -o pmc -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --no-graph --precision bf16 > $OUT/pmc_fetch_policy_bf16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_policy_bf16 -o pmc -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --no-graph --precision bf16 > $OUT/pmc_write_policy_bf16.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_video -o pmc -- python $R/tools/video_only.py --steps 1 --no-graph > $OUT/pmc_fetch_video.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_video -o pmc -- python $R/tools/video_only.py --steps 1 --no-graph > $OUT/pmc_write_video.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_video_bf16 -o pmc -- python $R/tools/video_only.py --steps 1 --no-graph --storage bf16 > $OUT/pmc_fetch_video_bf16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_video_bf16 -o pmc -- python $R/tools/video_only.py --steps 1 --no-graph --storage bf16 > $OUT/pmc_write_video_bf16.log 2>&1
# MFMA utilisation (matrix-pipe busy cycles vs active clock), one pass per leg
for leg in policy video video_bf16; do
  case $leg in
    policy) CMD="python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --no-graph";;
    video) CMD="python $R/tools/video_only.py --steps 1 --no-graph";;
    video_bf16) CMD="python $R/tools/video_only.py --steps 1 --storage bf16 --no-graph";;
  esac
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_$leg -o pmc -- $CMD > $OUT/pmc_mfma_$leg.log 2>&1
  python $R/tools/mfma_util.py $OUT/pmc_mfma_$leg > $OUT/mfma_util_$leg.json 2>> $OUT/pmc_summary.err
done
# keep only summaries small enough to travel back (<= 64 MiB total)
python $R/tools/summarize_pmc.py $OUT > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err
python -c "import json,sys; j=json.load(open('$OUT/pmc_summary.json')); json.dump(j['roofline_traffic'], open('$OUT/roofline_traffic.json','w'), indent=1)"
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT $OUT/* | head -60
grep -h '^{"metric' $OUT/policy_bench.log $OUT/video_bench.log | cut -c1-300
