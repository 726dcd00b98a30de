# round 5, first GPU call: RCCL on a one-rank group (test + bench line), all eight shares with and without the group
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/r05a/rccl_test.txt 2>&1; echo "rccl test rc $?" >> gpurun_out/r05a/rccl_test.txt
tail -30 gpurun_out/r05a/rccl_test.txt
timeout 600 python bench.py --force-dist --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 50 > gpurun_out/r05a/bench_force_dist.json 2> gpurun_out/r05a/bench_force_dist.err; tail -3 gpurun_out/r05a/bench_force_dist.err; cat gpurun_out/r05a/bench_force_dist.json
timeout 600 python bench.py --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 50 > gpurun_out/r05a/bench_plain.json 2>/dev/null; cat gpurun_out/r05a/bench_plain.json
P=8 timeout 1500 bash tools/emulate_shares_all.sh > gpurun_out/r05a/emulated_shares_all.txt 2>&1; cat gpurun_out/r05a/emulated_shares_all.txt
FORCE=1 P=8 timeout 1500 bash tools/emulate_shares_all.sh > gpurun_out/r05a/emulated_shares_all_rccl.txt 2>&1; cat gpurun_out/r05a/emulated_shares_all_rccl.txt
