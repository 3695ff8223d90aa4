# end-of-round validation: smoke, the whole GPU suite, profiles, the default bench line
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.txt 2>&1
timeout 2000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -4 > gpurun_out/final_tests.txt
bash tools/profile_round.sh r04c > gpurun_out/profile_r04c.log 2>&1
python bench.py > gpurun_out/final_bench_line.json 2> gpurun_out/final_bench.err; cp bench_full_last.json gpurun_out/final_bench_full.json
tail -2 gpurun_out/final_smoke.txt; cat gpurun_out/final_tests.txt; head -c 300 gpurun_out/final_bench_line.json
