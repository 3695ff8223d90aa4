set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r5a_tests.txt 2>&1
echo "tests exit $?" >> gpurun_out/r5a_tests.txt
tail -5 gpurun_out/r5a_tests.txt
python bench.py --steps 20 --warmup 5 --no-video --no-cpu-baseline --no-predict > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
tail -c 1500 gpurun_out/r5a_bench.json
