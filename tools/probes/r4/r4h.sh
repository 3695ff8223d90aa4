for i in 1 2 3; do for v in "V2A_UNET_WG_EARLY=0" "V2A_UNET_WG_EARLY=1"; do
  echo "== $v" >> gpurun_out/r4_early_step.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|unet_fwd begin|unet_bwd|unet_wgrad|enc_bwd.*(begin|chain done|end)|optimiser begin|step end|Error|error" >> gpurun_out/r4_early_step.txt
done; done
cat gpurun_out/r4_early_step.txt
python -m pytest tests/test_policy_gpu.py -x -q -m gpu -k "golden or reproducible or graph" 2>&1 | tail -2
V2A_UNET_WG_EARLY=1 python -m pytest tests/test_policy_gpu.py -x -q -m gpu -k "golden or reproducible or graph or three_train" 2>&1 | tail -2
