#!/bin/bash
cd $GRAFT_REPO_ROOT
WORKLOAD=C3 timeout -k 10 300 python scripts/ab.py "" "lds_min_kb_dw=100" "rows_per_tile=8,lds_budget_kb_dw=128" "rows_per_tile=8" "rows_per_tile=8,lds_min_kb_dw=100" 2>&1 | grep -v amdgpu.ids
