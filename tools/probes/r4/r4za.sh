timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -m gpu -k "three_train or graph or fp16" 2>&1 | grep -v Warning | tail -3
for i in 1 2 3; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-video --no-predict --no-roofline-pass --no-bf16-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('parked', d['ms_per_step'], d['final_loss'])"
done
V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | grep -E "optimiser|step end|ms per step"
