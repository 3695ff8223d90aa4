#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ...   -> policy steps/s (fp32) for each value of the environment variable
VAR=$1; shift
for v in "$@"; do
  r=$(env $VAR=$v python bench.py --steps 30 --warmup 5 --no-video --no-cpu-baseline --no-predict --no-bf16-extra --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],3))")
  echo "$VAR=$v -> $r"
done
