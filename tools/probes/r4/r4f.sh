python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_policy_gpu.py -x -q -m gpu -k "golden or reproducible or dp_gradient or three_train" 2>&1 | tail -3
for i in 1 2; do for v in "V2A_WGRAD_X3=1" "V2A_WGRAD_X3=0" "V2A_WGRAD_X3=1 V2A_DMA_SMALL_TILE=0" "V2A_WGRAD_X3=1 V2A_CONV_SLOTS=1024"; do
  echo "== $v" >> gpurun_out/r4_wgx3_step.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|unet_fwd begin|unet_bwd|enc_bwd.*(begin|chain done|end)|optimiser begin|step end" >> gpurun_out/r4_wgx3_step.txt
done; done
cat gpurun_out/r4_wgx3_step.txt
