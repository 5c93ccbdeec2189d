WORKLOAD=C5 python scripts/ab.py "" "debug=32" 2>&1 | grep -v amdgpu.ids
python scripts/ab.py "" "debug=32" 2>&1 | grep -v amdgpu.ids
