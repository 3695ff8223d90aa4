set -u
R=$PWD
mkdir -p gpurun_out/sb
cd $R
for rep in 1 2; do
  for tree in _r05 .; do
    (cd $R/$tree && python bench.py --steps 30 --warmup 5 --no-video --no-bf16-extra --no-predict --no-cpu-baseline --no-roofline-pass --no-video-train 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('policy', '$tree', d['ms_per_step'])")
  done
done
for tree in _r05 .; do
  (cd $R/$tree && python tools/video_only.py 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('video fp32', '$tree', round(d['value'],2), round(d['seconds_per_sample_call'],2))")
  (cd $R/$tree && python tools/video_only.py --storage bf16 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('video bf16', '$tree', round(d['value'],2))")
done
