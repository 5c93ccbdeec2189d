#!/bin/bash
# round-6 profile collection on one MI355X (PART=1|2|3): everything lands under gpurun_out/r06/, the summaries to keep are copied to profiles/ by hand.
# Collect AFTER the last kernel commit: bench.py quotes profiles/traffic.json and profiles/kernel_trace.json only for its own KERNELS_STAMP.
R=gpurun_out/r06; mkdir -p $R
export TMPDIR=/tmp
PART=${PART:-1}
STAMP=$(python3 -c "import re;print(re.search(r'KERNELS_STAMP = \"([^\"]+)\"', open('bench.py').read()).group(1))")
if [ "$PART" = "1" ]; then
  # the driver-shaped bench line, then the same bench under rocprofv3 (kernel stats of the PRODUCT alone: no CPU baseline / other workloads / solve / host-array / Lanczos legs --
  # the real-vector product launches the complex pass-B kernel on row pairs, which would mix into its average)
  timeout -k 10 600 python bench.py --steps 100 --warmup 20 2> $R/bench_err.log > $R/bench_n1.json; cut -c1-300 $R/bench_n1.json
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$R/bench_trace -- python3 /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --no-gf-solve --no-apply-host --no-lanczos > /root/repo/$R/bench_trace.log 2>&1 )
  cp $R/bench_trace/*/*kernel_stats.csv $R/bench_kernel_stats.csv 2>/dev/null; tail -1 $R/bench_trace.log | cut -c1-200
  python3 scripts/make_kernel_trace.py $R/bench_kernel_stats.csv $STAMP r06_bench_n1_kernel_stats.csv > $R/kernel_trace.log 2>&1; cp profiles/kernel_trace.json $R/; head -30 $R/kernel_trace.log
fi
if [ "$PART" = "2" ]; then
  # HBM-side traffic of the product at C3 / C4 / C5 (separate --pmc passes), SQ / LDS / TA counters at C3 (complex and real vectors)
  for W in C3 C4 C5; do
    WORKLOAD=$W bash scripts/prof_traffic.sh r06_$W "" > /dev/null 2>&1; cp gpurun_out/traffic_r06_$W/summary.txt $R/traffic_$W.txt
    python3 scripts/make_traffic.py $R/traffic_$W.txt $STAMP $W > $R/make_traffic_$W.log 2>&1; cat $R/traffic_$W.txt
  done
  cp profiles/traffic*.json $R/
  bash scripts/prof_sq.sh r06_c3 "" > /dev/null 2>&1; cp gpurun_out/sq_r06_c3/summary.txt $R/sq_C3.txt; head -50 $R/sq_C3.txt
  REAL=1 bash scripts/prof_sq.sh r06_c3_real "" > /dev/null 2>&1; cp gpurun_out/sq_r06_c3_real/summary.txt $R/sq_C3_real.txt
  REAL=1 bash scripts/prof_traffic.sh r06_c3_real "" > /dev/null 2>&1; cp gpurun_out/traffic_r06_c3_real/summary.txt $R/traffic_C3_real.txt; cat $R/traffic_C3_real.txt
fi
if [ "$PART" = "3" ]; then
  # hxv_eigh_lowest at C3 and C4: wall time, products, kernel budget
  for W in C3 C4; do
    ( cd /tmp && SECTOR=$W timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$R/eigh_$W -o eigh -- python3 /root/repo/scripts/eigh_c3.py > /root/repo/$R/eigh_$W.log 2>&1 )
    f=$(find $R/eigh_$W -name "*kernel_stats.csv" | head -1)
    python3 scripts/kernel_budget.py "$f" 24 >> $R/eigh_$W.log 2>&1
    cat $R/eigh_$W.log | grep -v "^W2\|amdgpu.ids" | head -40
  done
fi
