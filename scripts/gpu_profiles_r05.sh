#!/bin/bash
# round-5 profile collection on one MI355X (PART=1|2|3): everything lands under gpurun_out/r05/, the summaries to keep are copied to profiles/ by hand
R=gpurun_out/r05; mkdir -p $R
export TMPDIR=/tmp
PART=${PART:-1}
if [ "$PART" = "1" ]; then
  # the driver-shaped bench line, then the same bench under rocprofv3 (kernel stats; no CPU baseline / other workloads / solve leg)
  timeout -k 10 600 python bench.py --steps 100 --warmup 20 2> $R/bench_err.log > $R/bench_n1.json; cut -c1-400 $R/bench_n1.json
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$R/bench_trace -- python3 /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --no-gf-solve > /root/repo/$R/bench_trace.log 2>&1 )
  cp $R/bench_trace/*/*kernel_stats.csv $R/bench_kernel_stats.csv 2>/dev/null; tail -1 $R/bench_trace.log | cut -c1-200
fi
if [ "$PART" = "2" ]; then
  # counters the round-4 review asked for: HBM traffic + SQ / LDS / TA of the REAL-vector pair at C3, SQ / LDS / TA at C4
  REAL=1 bash scripts/prof_traffic.sh r05_c3_real "" > /dev/null 2>&1; cp gpurun_out/traffic_r05_c3_real/summary.txt $R/traffic_C3_real.txt
  REAL=1 bash scripts/prof_sq.sh r05_c3_real "" > /dev/null 2>&1; cp gpurun_out/sq_r05_c3_real/summary.txt $R/sq_C3_real.txt
  WORKLOAD=C4 bash scripts/prof_sq.sh r05_c4 "" > /dev/null 2>&1; cp gpurun_out/sq_r05_c4/summary.txt $R/sq_C4.txt
  cat $R/traffic_C3_real.txt; head -40 $R/sq_C3_real.txt; head -40 $R/sq_C4.txt
fi
if [ "$PART" = "3" ]; then
  # hxv_eigh_lowest at C3 and C4: wall time, products (search / check), kernel budget
  for W in C3 C4; do
    ( cd /tmp && SECTOR=$W timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$R/eigh_$W -o eigh -- python3 /root/repo/scripts/eigh_c3.py > /root/repo/$R/eigh_$W.log 2>&1 )
    f=$(find $R/eigh_$W -name "*kernel_stats.csv" | head -1)
    python3 scripts/kernel_budget.py "$f" 24 >> $R/eigh_$W.log 2>&1
    cat $R/eigh_$W.log | grep -v "^W2\|amdgpu.ids" | head -40
  done
fi
