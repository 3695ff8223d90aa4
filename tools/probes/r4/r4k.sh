for i in 1 2; do for v in "V2A_WGRAD_X3=1" "V2A_WGRAD_X3=0"; do
  echo "== $v" >> gpurun_out/r4_wgx3_step.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|unet_fwd begin|unet_bwd|enc_bwd.*img_obs_1.*(begin|chain done|end)|optimiser begin|step end|Error|error" >> gpurun_out/r4_wgx3_step.txt
done; done
cat gpurun_out/r4_wgx3_step.txt
