mkdir -p gpurun_out
python scripts/ab.py "" "threads_up=512,rows_per_thread_up=2" "cols_per_tile=2,threads_up=512,rows_per_thread_up=2" "cols_per_tile=2,rows_per_thread_up=2" "cols_per_tile=1,rows_per_thread_up=4" "cols_per_tile=2,rows_per_thread_up=4,lds_budget_kb_up=128" "cols_per_tile=4,rows_per_thread_up=2,lds_budget_kb_up=128" "cols_per_tile=1,threads_up=256,rows_per_thread_up=4" "cols_per_tile=2,threads_up=256,rows_per_thread_up=4" > gpurun_out/r03_mr_C3.log 2>&1
cat gpurun_out/r03_mr_C3.log
