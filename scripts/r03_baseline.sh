# round-3 baseline on today's box: shipped defaults + a sweep of the existing plan options (no code change)
mkdir -p gpurun_out
python scripts/ab.py "" "job_up=1" "rows_per_tile=8,lds_budget_kb_dw=128" "rows_per_tile=8" "pair_rows=1" > gpurun_out/r03_base_C3.log 2>&1
cat gpurun_out/r03_base_C3.log
WORKLOAD=C4 python scripts/ab.py "" "rows_per_tile=8,lds_budget_kb_dw=128" "rows_per_tile=8" > gpurun_out/r03_base_C4.log 2>&1
cat gpurun_out/r03_base_C4.log
WORKLOAD=C5 python scripts/ab.py "" "rows_per_tile=8,lds_budget_kb_dw=128" "rows_per_tile=4,pair_rows=1" "rows_per_tile=4,pair_rows=0" "cols_per_tile=2" > gpurun_out/r03_base_C5.log 2>&1
cat gpurun_out/r03_base_C5.log
