#!/bin/bash
# same-box A/B of environment settings on the policy step: r3_env_ab.sh "NAME=VAL" ...  (each compared with the default, fp32 and bf16)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do
for cfg in "X=0" "$@"; do
  for P in fp32 bf16; do
    env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass --precision $P 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $P', round(d['ms_per_step'],3), 'ms')"
  done
done
done
