python tools/probes/conv_stamp_probe.py > gpurun_out/r4_conv_stamps2.txt 2>&1
python tools/probes/conv_stamp_probe.py video > gpurun_out/r4_conv_stamps2_video.txt 2>&1
python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r4_ops_test.txt 2>&1; tail -5 gpurun_out/r4_ops_test.txt
for v in "V2A_F32P=0" "V2A_F32P=1" "V2A_F32P=1 V2A_F32P_S128=3 V2A_F32P_S64=3"; do
  echo "== $v" >> gpurun_out/r4_f32p_step.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|unet_fwd begin|unet_bwd|enc_bwd.*(begin|chain done|end)|optimiser begin" >> gpurun_out/r4_f32p_step.txt
done
cat gpurun_out/r4_conv_stamps2.txt gpurun_out/r4_conv_stamps2_video.txt gpurun_out/r4_f32p_step.txt
