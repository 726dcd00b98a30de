# round 5, second GPU call: the anomaly-mode tests, the gradient rule on the tests that used granted tolerances, the reference-shaped loop's
# host profile, all eight shares
mkdir -p gpurun_out/r05b
timeout 1500 python -m pytest tests/test_gpu_reference_loop.py -q -m gpu -x --durations=5 > gpurun_out/r05b/reference_loop_test.txt 2>&1; echo "rc $?" >> gpurun_out/r05b/reference_loop_test.txt
tail -40 gpurun_out/r05b/reference_loop_test.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_multirank.py -q -m gpu --durations=10 > gpurun_out/r05b/grad_rule_tests.txt 2>&1; echo "rc $?" >> gpurun_out/r05b/grad_rule_tests.txt
tail -60 gpurun_out/r05b/grad_rule_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1200 python tools/reference_loop_bench.py --profile > gpurun_out/r05b/reference_loop_bench.jsonl 2> gpurun_out/r05b/reference_loop_profile.txt; cat gpurun_out/r05b/reference_loop_bench.jsonl; cat gpurun_out/r05b/reference_loop_profile.txt | grep -v Warning | head -150
FORCE=1 P=8 timeout 1500 bash tools/emulate_shares_all.sh > gpurun_out/r05b/emulated_shares_all_rccl.txt 2>&1; cat gpurun_out/r05b/emulated_shares_all_rccl.txt
