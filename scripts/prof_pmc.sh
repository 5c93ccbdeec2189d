#!/bin/bash
# usage: prof_pmc.sh <tag> <cfg> ; collects kernel-trace stats + PMC passes into gpurun_out/prof_<tag>
set -e
TAG=$1; CFG=${2:-64,4,1024,64,4,1024,0}; KERN=${3:-1}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/scripts/prof_apply.py $CFG 3 $KERN > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 /root/repo/scripts/prof_apply.py $CFG 2 $KERN > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $OUT/errors.log
done
python3 /root/repo/scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
