# round-2 profiles: bench under rocprofv3 (kernel stats), traffic counters for C3/C4/C5, SQ counters for C3
mkdir -p gpurun_out/r02
python bench.py --steps 100 --warmup 20 2> gpurun_out/r02/bench_err.log > gpurun_out/r02/bench_n1_C3.json; cut -c1-400 gpurun_out/r02/bench_n1_C3.json
for wl in C4 C5; do python bench.py --steps 20 --warmup 5 --workload $wl --no-cpu-baseline 2>> gpurun_out/r02/bench_err.log > gpurun_out/r02/bench_n1_$wl.json; cut -c1-300 gpurun_out/r02/bench_n1_$wl.json; done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02/bench_trace -- python3 /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/r02/bench_trace.log 2>&1 )
cp gpurun_out/r02/bench_trace/*/*kernel_stats.csv gpurun_out/r02/bench_kernel_stats.csv 2>/dev/null
bash scripts/prof_traffic.sh r02_c3 "" > /dev/null 2>&1; cp gpurun_out/traffic_r02_c3/summary.txt gpurun_out/r02/traffic_C3.txt
WORKLOAD=C4 bash scripts/prof_traffic.sh r02_c4 "" > /dev/null 2>&1; cp gpurun_out/traffic_r02_c4/summary.txt gpurun_out/r02/traffic_C4.txt
WORKLOAD=C5 bash scripts/prof_traffic.sh r02_c5 "" > /dev/null 2>&1; cp gpurun_out/traffic_r02_c5/summary.txt gpurun_out/r02/traffic_C5.txt
bash scripts/prof_sq.sh r02_c3 "" > /dev/null 2>&1; cp gpurun_out/sq_r02_c3/summary.txt gpurun_out/r02/sq_C3.txt
WORKLOAD=C4 bash scripts/prof_sq.sh r02_c4 "" > /dev/null 2>&1; cp gpurun_out/sq_r02_c4/summary.txt gpurun_out/r02/sq_C4.txt
cat gpurun_out/r02/traffic_C3.txt gpurun_out/r02/traffic_C4.txt gpurun_out/r02/traffic_C5.txt
