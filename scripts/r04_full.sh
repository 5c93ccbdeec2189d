# full GPU check of a build: tests, smoke, bench (+cpu baseline)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_full.log 2>&1; tail -6 gpurun_out/r04_pytest_full.log
python __graft_entry__.py smoke 2>&1 | tail -1 &&
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; cut -c1-1500 gpurun_out/r04_bench_final.json
