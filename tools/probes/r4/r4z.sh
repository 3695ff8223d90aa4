for i in 1 2; do for v in "V2A_X=1" "V2A_WGRAD_MULTI_WG_X3H=768" "V2A_WGRAD_MULTI_WG_X3H=1024" "V2A_WGRAD_X3H_MIN_OW=16" "V2A_WGRAD_X3H_MIN_OW=32"; do
env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-video --no-predict --no-roofline-pass --no-bf16-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['ms_per_step'])"
done; done
