# round 5, tenth GPU call: forward / backward replayed inside the modules (gnan_amd.replay); reference-shaped loop again; whole suite
mkdir -p gpurun_out/r05j
timeout 900 python -m pytest tests/test_gpu_replay.py -q -m gpu -x 2>&1 | grep -v "Warning\|warn" | tail -30
timeout 900 python tools/reference_loop_bench.py > gpurun_out/r05j/reference_loop.jsonl 2>/dev/null; cat gpurun_out/r05j/reference_loop.jsonl | cut -c1-600
timeout 2400 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r05j/gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r05j/gpu_suite.txt; grep -v "Warning\|warn" gpurun_out/r05j/gpu_suite.txt | tail -40
