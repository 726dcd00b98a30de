# round 5, sixth GPU call: prefetched partial sums in the feature-sum look-up; padded parameter store (F = 129 through the fast kernels)
mkdir -p gpurun_out/r05f
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -q -m gpu -x -k "index or fpwl or lookup or look_up or table or pwl or moments or shape" 2>&1 | tail -4
timeout 900 python tools/lookup_ab.py 2>/dev/null | head -3 | cut -c1-400
timeout 900 python tools/train_step_c4.py 2>/dev/null | cut -c1-600
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_graphed.py tests/test_gpu_harness.py tests/test_gpu_reference_loop.py tests/test_gpu_parity.py -q -m gpu -x --durations=8 2>&1 | grep -v "Warning\|warn" | tail -25
bash tools/r04_c3tl.sh > gpurun_out/r05f/c3_timeline.txt 2>&1; grep -c dur gpurun_out/r05f/c3_timeline.txt; cat gpurun_out/r05f/c3_timeline.txt | tail -50 | cut -c1-150
timeout 600 python bench.py --config c3 --loop reference --no-cpu-baseline > gpurun_out/r05f/c3_loop_reference.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05f/c3_loop_reference.json')); print({k: d[k] for k in ('fwd_ms','fwd_bwd_ms','ms_per_step','replayed_fwd_bwd_ms','replay_note','reference_loop', 'stages_ms')})"
