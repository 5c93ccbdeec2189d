# end-of-round check of the final build: full GPU suite, smoke, bench line (with the stamped traffic), soaks, the N>1 rehearsals
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_full.log 2>&1; tail -4 gpurun_out/r04_pytest_full.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; cut -c1-300 gpurun_out/r04_bench_final.json; echo
timeout -k 10 300 python scripts/fuzz_soak.py 64 3000 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r04_fuzz_soak.log
timeout -k 10 300 python scripts/ranks_soak.py 2000 3500 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r04_ranks_soak.log
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29573 bench.py --gpus 2 --backend gloo --workload C2 --steps 3 --warmup 1 --parallelism sectors 2>&1 | grep "^{" | cut -c1-400
