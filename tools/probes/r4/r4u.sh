timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -40
for v in 1 0; do echo "== V2A_STEM_WINDOW=$v"; V2A_STEM_WINDOW=$v V2A_TSTAMP=1 python tools/phase_clock.py fp32 2>&1 | tail -30; done
