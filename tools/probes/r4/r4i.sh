python tools/probes/emu_probe.py 2>&1 | grep -E "f32x3\)|LDS-DMA" > gpurun_out/r4_emu_probe5.txt; cat gpurun_out/r4_emu_probe5.txt
python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do for v in "V2A_X3_NS64=4" "V2A_X3_NS64=2"; do
  echo "== $v" >> gpurun_out/r4_ns_step.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|unet_fwd begin|unet_bwd|enc_bwd.*(begin|chain done|end)|optimiser begin|step end|Error|error" >> gpurun_out/r4_ns_step.txt
done; done
grep -v "img_goal" gpurun_out/r4_ns_step.txt
