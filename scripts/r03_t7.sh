#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_lanczos.py tests/test_gpu_ranks.py -x -q -m gpu > gpurun_out/t7_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/t7_tests.log
tail -5 gpurun_out/t7_tests.log
timeout -k 10 200 python scripts/eigh_c3.py > gpurun_out/t7_eigh.log 2>&1
grep -v amdgpu.ids gpurun_out/t7_eigh.log
