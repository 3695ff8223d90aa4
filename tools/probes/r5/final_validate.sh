set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r5_final_tests.txt 2>&1
echo "tests exit $?" >> gpurun_out/r5_final_tests.txt
tail -6 gpurun_out/r5_final_tests.txt
python -m pytest tests/test_policy_gpu.py tests/test_joint_loop.py -m gpu -q -s -k "golden_and_oracle or batch_256 or ragged_batch or c2_batch64 or batched_exploration" 2>&1 | grep "^\[" > gpurun_out/r5_final_parity_lines.txt
cat gpurun_out/r5_final_parity_lines.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_final_smoke.txt 2>&1; tail -3 gpurun_out/r5_final_smoke.txt
python bench.py > gpurun_out/r5_final_bench_line.json 2> gpurun_out/r5_final_bench.err
cp bench_full_last.json gpurun_out/r5_final_bench_full.json 2>/dev/null
tail -c 600 gpurun_out/r5_final_bench_line.json
