python tools/probes/r4/unet_conv_time.py 2>&1 | grep "defer=1"
for i in 1 2 3; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-video --no-predict --no-roofline-pass --no-bf16-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('new plan', d['ms_per_step'], d['final_loss'])"
done
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -m gpu -k "golden or three_train or graph" 2>&1 | grep -v Warning | tail -3
