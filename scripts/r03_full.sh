# full GPU check of a build: tests, smoke, bench (+cpu baseline)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_full.log 2>&1; tail -6 gpurun_out/r03_pytest_full.log
python __graft_entry__.py smoke 2>&1 | tail -1 &&
python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; cut -c1-400 gpurun_out/r03_bench_final.json
