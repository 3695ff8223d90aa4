# state of the tree at session 3 of round 4: phase clock fp32 + bf16, short bench, bf16 sampler kernel stats
V2A_TSTAMP=1 python tools/phase_clock.py fp32 > gpurun_out/r4s_phase_fp32.txt 2>&1
V2A_TSTAMP=1 python tools/phase_clock.py bf16 > gpurun_out/r4s_phase_bf16.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-video --no-predict --no-bf16-extra 2>gpurun_out/r4s_bench.err | tail -1 > gpurun_out/r4s_bench.json
bash tools/run_policy_profile.sh fp32 r4s_pol > gpurun_out/r4s_timeline.txt 2>&1
cp gpurun_out/r4s_pol/*/*kernel_stats.csv gpurun_out/r4s_policy_kernel_stats.csv 2>/dev/null || find gpurun_out/r4s_pol -name "*stats*" 
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4s_vid -o video -- python $GRAFT_REPO_ROOT/tools/video_only.py --steps 3 --storage bf16 > $GRAFT_REPO_ROOT/gpurun_out/r4s_vid.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r4s_vid -name "*kernel_trace.csv" -delete; find gpurun_out/r4s_pol -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/r4s_vid.log; head -c 600 gpurun_out/r4s_bench.json
