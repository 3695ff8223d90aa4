timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "three_plane_halo" 2>&1 | tail -2
for v in 1 0 1 0; do V2A_CONV_X3H_64=$v python tools/video_only.py --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('x3h_64=$v', d['value'], d['seconds_per_sample_call'])"; done
