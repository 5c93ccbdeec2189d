mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_lanczos.py tests/test_gpu_ranks.py -m gpu -x -q -k "eigh or drivers" > gpurun_out/r03_eigh.log 2>&1; tail -8 gpurun_out/r03_eigh.log
python scripts/eigh_c3.py 2>&1 | grep -v amdgpu.ids | tail -5
