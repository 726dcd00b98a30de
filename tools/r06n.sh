python -m pytest tests/test_gpu_kernels.py -q -x -k "bucketed_pairs_by_the_library or propagation_blocked" 2>&1 | tail -4
python tools/pb_bench.py 1 2 2>&1 | tail -2 | cut -c1-300

python bench.py --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 10 --order sum_first 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sum_first ms', round(d['ms_per_step'],4), d['amortised_setup_ms'], d['checksum'])"
python tools/train_step_c4.py 2>&1 | tail -1
PER_LAYER=1 python tools/train_step_c4.py 2>&1 | tail -1
