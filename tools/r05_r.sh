mkdir -p gpurun_out/r05r
timeout 900 python -m pytest tests/test_gpu_replay.py tests/test_gpu_graphed.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python bench.py --config c2 --steps 3 --warmup 1 > gpurun_out/r05r/c2_bench.log 2>&1; grep "^{" gpurun_out/r05r/c2_bench.log | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('ms_per_graph','replayed_eval_ms_per_graph','replayed_train_ms_per_graph','kernels_per_replayed_training_step','captured_training_steps')}, d['cpu_baseline']['parity_max_rel_err'])"
timeout 900 python bench.py --config c2 --loop reference --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r05r/c2_loop_reference.log 2>&1; grep "^{" gpurun_out/r05r/c2_loop_reference.log | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print(d['reference_loop'])"; tail -3 gpurun_out/r05r/c2_loop_reference.log | cut -c1-300
