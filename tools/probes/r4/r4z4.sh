timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "groupnorm" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_policy_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -3
for i in 1 2 3; do for v in "V2A_GN_WAVEV=1" "V2A_GN_WAVEV=0"; do
env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-video --no-predict --no-roofline-pass --no-bf16-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['final_loss'])"
done; done
