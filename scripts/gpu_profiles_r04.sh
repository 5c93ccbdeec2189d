# round-3 profiles: bench line, bench under rocprofv3 (kernel stats), HBM-traffic counters for C3/C4/C5, SQ counters for C3
R=gpurun_out/r04; mkdir -p $R
python bench.py --steps 100 --warmup 20 2> $R/bench_err.log > $R/bench_n1.json; cut -c1-600 $R/bench_n1.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$R/bench_trace -- python3 /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads > /root/repo/$R/bench_trace.log 2>&1 )
cp $R/bench_trace/*/*kernel_stats.csv $R/bench_kernel_stats.csv 2>/dev/null; tail -1 $R/bench_trace.log | cut -c1-300
bash scripts/prof_traffic.sh r04_c3 "" > /dev/null 2>&1; cp gpurun_out/traffic_r04_c3/summary.txt $R/traffic_C3.txt
WORKLOAD=C4 bash scripts/prof_traffic.sh r04_c4 "" > /dev/null 2>&1; cp gpurun_out/traffic_r04_c4/summary.txt $R/traffic_C4.txt
WORKLOAD=C5 bash scripts/prof_traffic.sh r04_c5 "" > /dev/null 2>&1; cp gpurun_out/traffic_r04_c5/summary.txt $R/traffic_C5.txt
bash scripts/prof_sq.sh r04_c3 "" > /dev/null 2>&1; cp gpurun_out/sq_r04_c3/summary.txt $R/sq_C3.txt
cat $R/traffic_C3.txt $R/traffic_C4.txt $R/traffic_C5.txt; head -30 $R/sq_C3.txt
