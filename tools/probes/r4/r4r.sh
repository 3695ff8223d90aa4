run() { env "$@" MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-video --no-predict --no-roofline-pass --no-bf16-extra 2>&1 >/dev/null | grep "bench full record" | python -c "import sys,json; d=json.loads(sys.stdin.read().split('] ',1)[1]); print(repr(d['final_loss']), d['value'])"; }
echo "single split=1"; run V2A_SPLIT_PACKS=1
for i in 1 2 3 4; do echo "dp split=1"; run V2A_SPLIT_PACKS=1 V2A_FORCE_DP=1; done
echo "dp split=0"; run V2A_SPLIT_PACKS=0 V2A_FORCE_DP=1
echo "single split=0"; run V2A_SPLIT_PACKS=0
