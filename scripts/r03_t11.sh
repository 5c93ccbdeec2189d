#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_lanczos.py tests/test_gpu_ranks.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/t11_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/t11_tests.log
tail -4 gpurun_out/t11_tests.log
