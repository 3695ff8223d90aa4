timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -m gpu -k "optimiser_written or bf16 or fp16 or three_train or graph" 2>&1 | grep -v Warning | tail -4
for i in 1 2 3; do for v in 1 0; do
V2A_FUSE_PACKS=$v python bench.py --steps 20 --warmup 5 --precision bf16 --no-cpu-baseline --no-video --no-predict --no-roofline-pass --no-bf16-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bf16 fuse=$v', d['ms_per_step'], d['final_loss'])"
done; done
