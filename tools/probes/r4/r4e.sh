for m in exact x3; do
  echo "== $m" >> gpurun_out/r4_ragged.txt
  V2A_F32_CONV=$m python tools/probes/ragged_grad_probe.py 7:0 7:1 7:2 7:3 7:4 7:5 7:6 7:7 6:0 6:1 6:2 6:3 6:4 6:5 2>&1 | grep "^B=" >> gpurun_out/r4_ragged.txt
done
cat gpurun_out/r4_ragged.txt
python -m pytest tests/test_video_gpu.py -q -m gpu -k "graph_replay_equals_eager or 16bit_samplers_drift or hipgraph_replay" -s 2>&1 | grep -E "50-step|passed|failed|Error|assert" | head
