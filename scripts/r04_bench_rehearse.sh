#!/bin/bash
# the N>1 script paths on a one-GPU box: one-rank C-ABI rehearsal of every exchange, and a two-rank gloo rehearsal of the torch twin
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_comm.py -x -q > gpurun_out/r04_rehearse_pytest.log 2>&1; tail -40 gpurun_out/r04_rehearse_pytest.log
for ex in allgather halo alltoall; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --backend gloo --workload C2 --steps 3 --warmup 1 --exchange $ex > gpurun_out/r04_rehearse_gloo_$ex.log 2>&1
  grep -v "amdgpu.ids\|Setting OMP" gpurun_out/r04_rehearse_gloo_$ex.log | grep -B2 -A12 "Traceback\|^{" | head -40 | cut -c1-600
done
