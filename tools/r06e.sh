OUT=gpurun_out/r06e; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -q -x -k "propagation_blocked" 2>&1 | tail -15
python tools/train_step_c4.py 2>&1 | tail -1
PB_OFF=1 python tools/train_step_c4.py 2>&1 | tail -1
python bench.py --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 20 --order sum_first 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sum_first ms', round(d['ms_per_step'],4), d['stages_ms'], d['checksum'], d['amortised_setup_ms'])"
python -m pytest tests/test_gpu_fullsize.py -q -x 2>&1 | tail -5
