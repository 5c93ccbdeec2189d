mkdir -p gpurun_out
python bench.py --steps 100 --warmup 20 2> gpurun_out/r03_bench_err.log | tee gpurun_out/r03_bench_n1.json | cut -c1-3000
tail -3 gpurun_out/r03_bench_err.log
