# round 5: group-split feature sum on medium batches
mkdir -p gpurun_out/r05o
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_replay.py -q -m gpu -x -k "index or fpwl or lookup or look_up or table or pwl or moments or shape or c3 or replay" 2>&1 | tail -5
bash tools/c3_timeline.sh > gpurun_out/r05o/c3_timeline.txt 2>&1; grep -c dur gpurun_out/r05o/c3_timeline.txt; grep "fpwl_index\|sum_groups\|moments\|total" gpurun_out/r05o/c3_timeline.txt | cut -c1-150
timeout 600 python bench.py --config c3 --loop reference --no-cpu-baseline > gpurun_out/r05o/c3.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05o/c3.json')); print({k: d[k] for k in ('fwd_ms','fwd_bwd_ms','ms_per_step','replayed_fwd_bwd_ms','replay_note','reference_loop')})"
