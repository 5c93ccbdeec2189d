#!/bin/bash
# usage: prof_stats.sh <tag> "<options>" [WORKLOAD] : rocprofv3 kernel stats of a few products -> gpurun_out/stats_<tag>/summary.txt
TAG=$1; OPTS=$2; export WORKLOAD=${3:-C3}
OUT=/root/repo/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/scripts/prof_opts.py "$OPTS" 5 > $OUT/trace.log 2>&1
python3 /root/repo/scripts/prof_summary2.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
