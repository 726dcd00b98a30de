mkdir -p gpurun_out/r05s
timeout 2400 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r05s/gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r05s/gpu_suite.txt; grep -v "Warning\|warn" gpurun_out/r05s/gpu_suite.txt | tail -25
