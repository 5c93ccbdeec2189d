mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lanczos.py -m gpu -x -q -k "paired or fused or graph or real_vector" > gpurun_out/r03_pair_tests.log 2>&1; tail -5 gpurun_out/r03_pair_tests.log
python scripts/pair_c3.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_pair_c3.log
