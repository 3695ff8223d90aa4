python tools/probes/emu_probe.py > gpurun_out/r4_emu_probe4.txt 2>&1
grep -E "f32x3|LDS-DMA" gpurun_out/r4_emu_probe4.txt
python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_video_gpu.py -q -m gpu -k "c3_full_size or bitwise or unet_libero_forward" -s 2>&1 | grep -E "C3|passed|failed|Error" | head
python -m pytest tests/test_policy_gpu.py -q -m gpu -k "ragged_batch or c2_batch64 or dp_gradient" -s 2>&1 | grep -E "ragged B|C2 B|dp-equality|passed|failed|Error|assert" | head -20
bash tools/run_policy_profile.sh fp32 r4_prof_x3 > gpurun_out/r4_prof_x3_timeline.txt 2>&1
for i in 1 2; do for v in "V2A_X3_BIG=1" "V2A_X3_BIG=0" "V2A_F32_CONV=exact"; do
  echo "== $v" >> gpurun_out/r4_x3_step2.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|unet_fwd begin|unet_bwd|enc_bwd.*(begin|chain done|end)|optimiser begin|step end" >> gpurun_out/r4_x3_step2.txt
done; done
cat gpurun_out/r4_x3_step2.txt
