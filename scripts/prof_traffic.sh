#!/bin/bash
# usage: prof_traffic.sh <tag> "<options k=v,...>" : kernel stats + the three HBM-traffic PMC passes -> gpurun_out/traffic_<tag>/summary.txt
TAG=$1; OPTS=$2
OUT=/root/repo/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/scripts/prof_opts.py "$OPTS" 5 > $OUT/trace.log 2>&1
i=0
for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 /root/repo/scripts/prof_opts.py "$OPTS" 2 > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $OUT/errors.log
done
python3 /root/repo/scripts/prof_summary2.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
