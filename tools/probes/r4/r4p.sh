for i in 1 2; do for v in "V2A_SPLIT_PACKS=1" "V2A_SPLIT_PACKS=0"; do
  echo "== $v" >> gpurun_out/r4_splitpack.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|unet_fwd begin|unet_bwd begin|enc_bwd.*img_obs_1.*(begin|chain done|end)|optimiser begin|packs begin|step end|Error|error" >> gpurun_out/r4_splitpack.txt
done; done
cat gpurun_out/r4_splitpack.txt
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -5
