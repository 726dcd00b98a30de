# round 5, fourth GPU call: C3 replay timeline (before merging kernels), host phases of the reference-shaped loop, --loop reference lines
mkdir -p gpurun_out/r05d
bash tools/r04_c3tl.sh > gpurun_out/r05d/c3_timeline_before.txt 2>&1; cat gpurun_out/r05d/c3_timeline_before.txt | tail -60
timeout 900 python tools/reference_loop_bench.py > gpurun_out/r05d/reference_loop_bench.jsonl 2> /dev/null; cat gpurun_out/r05d/reference_loop_bench.jsonl | cut -c1-700
timeout 600 python bench.py --config c3 --loop reference --no-cpu-baseline > gpurun_out/r05d/c3_loop_reference.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05d/c3_loop_reference.json')); print({k: d[k] for k in ('fwd_ms','fwd_bwd_ms','ms_per_step','replayed_fwd_bwd_ms','replay_note','reference_loop')})"
timeout 900 python bench.py --config c2 --loop reference --no-cpu-baseline > gpurun_out/r05d/c2_loop_reference.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05d/c2_loop_reference.json')); print({k: d[k] for k in ('ms_per_graph','replayed_eval_ms_per_graph','replayed_train_ms_per_graph','reference_loop')})"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "integration_stub or uncapturable or empty_graph" 2>&1 | tail -15
