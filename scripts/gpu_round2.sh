# full GPU check of a build: tests, smoke, bench (+cpu baseline), N>1 rehearsals of bench on one GPU (gloo), rocprof kernel stats
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/pytest_full.log 2>&1; tail -6 gpurun_out/pytest_full.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --steps 100 --warmup 20 2> gpurun_out/bench_err.log | tee gpurun_out/bench_n1.json | cut -c1-1500
for ex in allgather alltoall halo; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 3 --backend gloo --exchange $ex --check --steps 3 --warmup 1 --workload C2 --no-lanczos > gpurun_out/bench_rehearsal_$ex.log 2>&1; tail -4 gpurun_out/bench_rehearsal_$ex.log | cut -c1-600
done
