mkdir -p gpurun_out
WORKLOAD=C5 python scripts/ab.py "" "cols_per_tile=2,rows_per_thread_up=2" "cols_per_tile=1,rows_per_thread_up=4" "cols_per_tile=2,rows_per_thread_up=4,lds_budget_kb_up=128" "cols_per_tile=4,rows_per_thread_up=2,lds_budget_kb_up=128" > gpurun_out/r03_mr_C5.log 2>&1
cat gpurun_out/r03_mr_C5.log
WORKLOAD=C4 python scripts/ab.py "" "cols_per_tile=2,rows_per_thread_up=2" "cols_per_tile=4,rows_per_thread_up=2,lds_budget_kb_up=128" > gpurun_out/r03_mr_C4.log 2>&1
cat gpurun_out/r03_mr_C4.log
