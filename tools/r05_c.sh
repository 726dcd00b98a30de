# round 5, third GPU call: the whole GPU suite with flat parameters() + the reference-shaped loop again
mkdir -p gpurun_out/r05c
timeout 2400 python -m pytest tests -q -m gpu --durations=25 -x > gpurun_out/r05c/gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r05c/gpu_suite.txt
grep -v "Warning\|warn" gpurun_out/r05c/gpu_suite.txt | tail -60
timeout 1200 python tools/reference_loop_bench.py --profile > gpurun_out/r05c/reference_loop_bench.jsonl 2> gpurun_out/r05c/reference_loop_profile.txt; cat gpurun_out/r05c/reference_loop_bench.jsonl; grep -v Warning gpurun_out/r05c/reference_loop_profile.txt | head -130
