set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_video_gpu.py tests/test_ops_gpu.py tests/test_joint_loop.py -m gpu -q --maxfail=8 > gpurun_out/r5h_tests.txt 2>&1
tail -12 gpurun_out/r5h_tests.txt
python -m pytest tests/test_policy_gpu.py -m gpu -q -k "presplit" >> gpurun_out/r5h_tests.txt 2>&1
tail -3 gpurun_out/r5h_tests.txt
python tools/video_only.py --steps 50 --storage bf16 > gpurun_out/r5h_video_bf16.json 2> gpurun_out/r5h_video_bf16.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5h_video_bf16.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "seconds_per_sample_call") if k in d}, d.get("roofline", {}).get("end_to_end_frac"))
PY
