# full GPU check of a build: tests, smoke, bench (+cpu baseline), rocprof kernel stats
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --steps 100 --warmup 20 2> gpurun_out/bench_err.log | tee gpurun_out/bench_n1.json | cut -c1-900
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/bench_trace -- python3 /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/bench_trace.log 2>&1
python3 /root/repo/scripts/prof_summary.py /root/repo/gpurun_out/bench_trace 2>&1 | cut -c1-330 | grep -E "pass_|naive|lz_" 
