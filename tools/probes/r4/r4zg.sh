timeout 1500 python -m pytest tests/test_policy_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -3
for i in 1 2 3; do for v in 1 0; do
V2A_PRESUM=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-video --no-predict --no-roofline-pass --no-bf16-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('presum=$v', d['ms_per_step'], d['final_loss'])"
done; done
