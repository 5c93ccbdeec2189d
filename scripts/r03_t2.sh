python scripts/ab.py "" "split_dw=1" "split_dw=1,threads_dw=512" 2>&1 | grep -v amdgpu.ids
WORKLOAD=C5 python scripts/ab.py "" "split_dw=1" 2>&1 | grep -v amdgpu.ids
