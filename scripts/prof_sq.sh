#!/bin/bash
# usage: prof_sq.sh <tag> "<options>" : SQ / LDS / TCP counters of the product kernels -> gpurun_out/sq_<tag>/summary.txt
TAG=$1; OPTS=$2
OUT=/root/repo/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 /root/repo/scripts/prof_opts.py "$OPTS" 2 > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $OUT/errors.log
done
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = row.get("Kernel_Name", "?")
        s = next((k for k in ("pass_up", "pass_dw", "up_job", "dw_job") if k in n), None)
        if s: agg[s][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in agg:
    print("==", k)
    for c in sorted(agg[k]):
        v = agg[k][c]; print(f"  {c:32s} {sum(v)/len(v):.4g}")
PY
cat $OUT/summary.txt
