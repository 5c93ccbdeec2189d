python scripts/lz_ab2.py "" "spread_banks=0" "" 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lanczos.py -m gpu -x -q 2>&1 | tail -2
