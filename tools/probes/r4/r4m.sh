timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -m gpu -k "dp_step_structure" 2>&1 | grep -v Warning | tail -40
for v in "256 128" "384 256" "512 128" "256 256" "768 256"; do set -- $v
  echo "== WG_X3_64=$1 WG_X3_128=$2" >> gpurun_out/r4_wgx3_tune2.txt
  V2A_WGRAD_MULTI_WG_X3_64=$1 V2A_WGRAD_MULTI_WG_X3_128=$2 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|Error|error" >> gpurun_out/r4_wgx3_tune2.txt
done
cat gpurun_out/r4_wgx3_tune2.txt
