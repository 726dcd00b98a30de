# round 5, fifth GPU call: anchor pairs in the direct-index look-up: bit-identity tests, A/B timing, SQ counters, training step
mkdir -p gpurun_out/r05e
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -q -m gpu -x -k "index or fpwl or lookup or look_up or table or pwl or moments or shape" 2>&1 | tail -8
timeout 900 python tools/lookup_ab.py > gpurun_out/r05e/lookup_ab.jsonl 2>/dev/null; cat gpurun_out/r05e/lookup_ab.jsonl | cut -c1-400
timeout 900 python tools/train_step_c4.py > gpurun_out/r05e/train_step_c4.json 2>/dev/null; cat gpurun_out/r05e/train_step_c4.json | cut -c1-1500
bash tools/pmc_sq_cmd.sh gpurun_out/r05e/sq_train python3 tools/train_step_c4.py > gpurun_out/r05e/sq_train.txt 2>&1; cat gpurun_out/r05e/sq_train.txt
timeout 600 python bench.py --order sum_first --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 20 > gpurun_out/r05e/sum_first_bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05e/sum_first_bench.json')); print('sum_first', d['ms_per_step'], d['stages_ms'])"
timeout 600 python bench.py --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 20 > gpurun_out/r05e/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05e/bench.json')); print('reference order', d['ms_per_step'], d['stages_ms'], d['amortised_setup_ms'])"
