mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_parity.log 2>&1; tail -3 gpurun_out/r03_parity.log
python scripts/ab.py "" "debug=64" "" "debug=64" > gpurun_out/r03_rs16_C3.log 2>&1; cat gpurun_out/r03_rs16_C3.log
WORKLOAD=C4 python scripts/ab.py "" "debug=64" > gpurun_out/r03_rs16_C4.log 2>&1; cat gpurun_out/r03_rs16_C4.log
WORKLOAD=C5 python scripts/ab.py "" "debug=64" > gpurun_out/r03_rs16_C5.log 2>&1; cat gpurun_out/r03_rs16_C5.log
