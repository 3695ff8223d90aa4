bash tools/run_policy_profile.sh fp32 r4x_pol > gpurun_out/r4x_timeline.txt 2>&1
V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | tail -28 > gpurun_out/r4x_phase.txt
