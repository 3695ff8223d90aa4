for i in 1 2; do for v in "V2A_X=1" "V2A_WGRAD_MULTI_WG_X3_128=256" "V2A_WGRAD_MULTI_WG_X3_128=1024" "V2A_WGRAD_MULTI_WG_X3_64=512" "V2A_WGRAD_MULTI_WG_X3_64=2048" "V2A_WGRAD_MULTI_DEPTH=8" "V2A_CONV_SLOTS=768"; do
env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-video --no-predict --no-roofline-pass --no-bf16-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['ms_per_step'])"
done; done
