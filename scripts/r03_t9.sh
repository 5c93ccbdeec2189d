#!/bin/bash
cd $GRAFT_REPO_ROOT
B="rows_per_tile=8,lds_budget_kb_dw=128,dw_loop=1"
WORKLOAD=C3 timeout -k 10 400 python scripts/ab.py "" "$B" "$B,debug=256" "$B,debug=512" "$B,debug=768" "$B,debug=1280" "$B,debug=2304" "$B,debug=3328" "$B,debug=3840" 2>&1 | grep -v amdgpu.ids
