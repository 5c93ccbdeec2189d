#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "kanamori or nonlocal or fuzz or Jx or nd" > gpurun_out/t10_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/t10_tests.log
tail -4 gpurun_out/t10_tests.log
WORKLOAD=C4K timeout -k 10 300 python scripts/ab.py "" "fold_nd=0" 2>&1 | grep -v amdgpu.ids
