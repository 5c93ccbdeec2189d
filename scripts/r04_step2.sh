#!/bin/bash
cd $GRAFT_REPO_ROOT
export HXV_EXPERIMENTS=1
{
for d in build_dbg/wt_a3bb891 . build_dbg/v1 build_dbg/v2; do
  timeout -k 10 200 python scripts/bisect_real.py $d 2>&1 | grep -v amdgpu.ids
done
timeout -k 10 200 python scripts/bisect_real.py . debug=64 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r04_bisect2.log
WORKLOAD=C5 timeout -k 10 600 python scripts/ab.py "" "block_order=1" "block_order=1,rows_per_tile=4" "block_order=1,lds_min_kb_dw=90" "block_order=1,lds_min_kb_up=90" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ab_c5_more.log
WORKLOAD=C5 bash scripts/prof_traffic.sh r04_c5_bo1 "block_order=1" > /dev/null 2>&1; cp gpurun_out/traffic_r04_c5_bo1/summary.txt gpurun_out/r04_traffic_C5_bo1.txt; cat gpurun_out/r04_traffic_C5_bo1.txt
