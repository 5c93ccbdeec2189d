#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lanczos.py -x -q -m gpu -k "spH0nd or fused_and_plain" > gpurun_out/t12_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/t12_tests.log
tail -12 gpurun_out/t12_tests.log
