# the lines gnan_amd.replay touches, once more after its last change (copies of static outputs, copyable models, plans for 129+ nodes)
mkdir -p gpurun_out/replay_lines
python3 tools/reference_loop_bench.py --profile > gpurun_out/replay_lines/reference_loop.jsonl 2> gpurun_out/replay_lines/reference_loop_host_profile.txt
python3 bench.py --config c3 > gpurun_out/replay_lines/c3_bench.log 2>&1
python3 bench.py --config c3 --loop reference --no-cpu-baseline > gpurun_out/replay_lines/c3_loop_reference.log 2>&1
python3 bench.py --config c2 --steps 3 --warmup 1 > gpurun_out/replay_lines/c2_bench.log 2>&1
python3 bench.py --config c2 --loop reference --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/replay_lines/c2_loop_reference.log 2>&1
python3 tools/graphed_step.py arxiv cora muta arxiv40 > gpurun_out/replay_lines/graphed_steps.log 2>&1
python3 bench.py > gpurun_out/replay_lines/bench_full.log 2>&1
cat gpurun_out/replay_lines/reference_loop.jsonl | cut -c1-300
