V2A_WGRAD_X3=1 timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "grouped_weight_gradients or wgrad" 2>&1 | tail -15
timeout 300 python tools/probes/wgrad_x3_probe.py 2>&1 | tee gpurun_out/r4_wgrad_probe.txt
V2A_WGRAD_X3=1 timeout 300 python tools/probes/wgrad_x3_probe.py 2>&1 | tee -a gpurun_out/r4_wgrad_probe.txt
