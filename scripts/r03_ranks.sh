mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ranks.py tests/test_gpu_comm.py -m gpu -x -q > gpurun_out/r03_ranks.log 2>&1; tail -15 gpurun_out/r03_ranks.log
