timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
python scripts/ab.py "" "sort_outer_dw=0" "" "sort_outer_dw=0" 2>&1 | grep -v amdgpu.ids
WORKLOAD=C4 python scripts/ab.py "" "sort_outer_dw=0" 2>&1 | grep -v amdgpu.ids
WORKLOAD=C5 python scripts/ab.py "" "sort_outer_dw=0" 2>&1 | grep -v amdgpu.ids
