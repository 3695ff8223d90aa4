python -m pytest tests/test_policy_gpu.py -x -q -m gpu -s 2>&1 | grep -E "ragged|C2 B|dp-equality|B=256|passed|failed|Error|assert|error" | head -30
for i in 1 2; do for v in "V2A_ENC_STACK=1" "V2A_ENC_STACK=0"; do
  echo "== $v" >> gpurun_out/r4_stack_step.txt
  env $v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|unet_fwd begin|unet_bwd|enc_bwd.*(begin|chain done|end)|enc_fwd.*end|optimiser begin|step end|Error|error" >> gpurun_out/r4_stack_step.txt
done; done
cat gpurun_out/r4_stack_step.txt
