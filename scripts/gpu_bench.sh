set -e
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_err.log | tee gpurun_out/bench_n1.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>> gpurun_out/bench_err.log | tee gpurun_out/bench_torchrun1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/bench_trace -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/bench_trace.log 2>&1
python3 /root/repo/scripts/prof_summary.py /root/repo/gpurun_out/bench_trace 2>&1 | cut -c1-400 | grep -v "at::native" 
