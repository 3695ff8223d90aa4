timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_policy_gpu.py -x -q -m gpu 2>&1 | tail -8
for v in "512 256" "1024 512" "2048 512" "1024 1024" "2048 1024" "4096 1024"; do set -- $v
  echo "== WG_X3_64=$1 WG_X3_128=$2" >> gpurun_out/r4_wgx3_tune.txt
  V2A_WGRAD_MULTI_WG_X3_64=$1 V2A_WGRAD_MULTI_WG_X3_128=$2 python tools/phase_clock.py fp32 2>&1 | grep -E "ms per step|Error|error" >> gpurun_out/r4_wgx3_tune.txt
done
cat gpurun_out/r4_wgx3_tune.txt
