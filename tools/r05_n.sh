mkdir -p gpurun_out/r05n
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r05n/gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r05n/gpu_suite.txt; grep -v "Warning\|warn" gpurun_out/r05n/gpu_suite.txt | tail -30
nproc
