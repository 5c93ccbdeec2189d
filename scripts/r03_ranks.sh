# thread-rank and one-rank RCCL tests (N>1 code of the C-ABI on one GPU)
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ranks.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/ranks_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/ranks_tests.log
tail -25 gpurun_out/ranks_tests.log
