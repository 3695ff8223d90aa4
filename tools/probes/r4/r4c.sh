python tools/probes/emu_probe.py > gpurun_out/r4_emu_probe2.txt 2>&1
python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r4_ops_test2.txt 2>&1; tail -15 gpurun_out/r4_ops_test2.txt
for v in "V2A_F32_CONV=exact" "V2A_F32_CONV=x3"; do
  echo "== $v" >> gpurun_out/r4_x3_step.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|unet_fwd begin|unet_bwd|enc_bwd.*(begin|chain done|end)|optimiser begin" >> gpurun_out/r4_x3_step.txt
done
cat gpurun_out/r4_emu_probe2.txt gpurun_out/r4_x3_step.txt
