python tools/probes/conv_stamp_probe.py > gpurun_out/r4_conv_stamps.txt 2>&1
V2A_TSTAMP=2 python tools/phase_clock.py fp32 > gpurun_out/r4_phase_fp32_fine.txt 2>&1
V2A_TSTAMP=2 python tools/phase_clock.py bf16 > gpurun_out/r4_phase_bf16_fine.txt 2>&1
for v in "" "V2A_PRIO=1" "V2A_WGRAD_BATCH=stage V2A_WGRAD_BATCH_STREAM=1" "V2A_PRIO=1 V2A_WGRAD_BATCH=stage V2A_WGRAD_BATCH_STREAM=1" "V2A_FLIP_DGRAD=0"; do
  echo "== $v" >> gpurun_out/r4_prio.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|enc_bwd|unet_wgrad|optimiser begin" >> gpurun_out/r4_prio.txt
done
cat gpurun_out/r4_conv_stamps.txt
