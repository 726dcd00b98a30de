# round 5, seventh GPU call: graph slots (one captured step per model for batch-size-1 graph tasks), moments block model at medium sizes
mkdir -p gpurun_out/r05g
timeout 1800 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_harness.py -q -m gpu -x --durations=6 2>&1 | grep -v "Warning\|warn" | tail -25
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "batched or moments or small" 2>&1 | tail -4
timeout 900 python tools/muta_epoch.py > gpurun_out/r05g/muta_epoch.json 2>gpurun_out/r05g/muta_epoch.err; tail -3 gpurun_out/r05g/muta_epoch.err; cat gpurun_out/r05g/muta_epoch.json | cut -c1-1500
bash tools/r04_c3tl.sh > gpurun_out/r05g/c3_timeline.txt 2>&1; grep "moments\|fpwl_index\|total" gpurun_out/r05g/c3_timeline.txt | cut -c1-150
timeout 600 python bench.py --config c3 --no-cpu-baseline > gpurun_out/r05g/c3.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05g/c3.json')); print({k: d[k] for k in ('fwd_ms','fwd_bwd_ms','ms_per_step','replayed_fwd_bwd_ms','replay_note')})"
timeout 900 python bench.py --config c2 --no-cpu-baseline > gpurun_out/r05g/c2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05g/c2.json')); print({k: d[k] for k in ('ms_per_graph','replayed_eval_ms_per_graph','replayed_train_ms_per_graph','kernels_per_replayed_training_step','captured_training_steps')})"
