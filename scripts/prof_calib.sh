#!/bin/bash
TAG=$1; CFG=${2:-64,4,1024,64,4,1024,0}
OUT=/root/repo/gpurun_out/calib_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_READ_sum TCC_READ_SECTORS_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_WRITE_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 /root/repo/scripts/prof_calib.py $CFG > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $OUT/errors.log
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        short = "pass_up" if "pass_up" in n else "pass_dw" if "pass_dw" in n else "copy" if ("copy" in n.lower() or "elementwise" in n) and "normal" not in n else None
        if short: agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in agg:
    print("==", k)
    for c in sorted(agg[k]):
        v = agg[k][c]; print(f"  {c:32s} {sum(v)/len(v):.4g} (n={len(v)})")
PY
