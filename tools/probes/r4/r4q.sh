timeout 600 python -m pytest tests/test_policy_gpu.py -x -q -m gpu -k dp_step_structure 2>&1 | grep -v Warning | tail -30
