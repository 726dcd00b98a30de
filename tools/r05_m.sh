# round 5: small-graph slot replay inside the modules; reference loop; whole suite
mkdir -p gpurun_out/r05m
timeout 900 python -m pytest tests/test_gpu_replay.py -q -m gpu -x 2>&1 | grep -v "Warning\|warn" | tail -30
timeout 900 python tools/reference_loop_bench.py muta > gpurun_out/r05m/reference_loop_muta.jsonl 2>/dev/null; cat gpurun_out/r05m/reference_loop_muta.jsonl | cut -c1-600
timeout 2400 python -m pytest tests -q -m gpu -x --durations=6 > gpurun_out/r05m/gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r05m/gpu_suite.txt; grep -v "Warning\|warn" gpurun_out/r05m/gpu_suite.txt | tail -40
