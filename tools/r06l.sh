OUT=gpurun_out/r06l; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -q -x -k "degree_sorted_copy_by_the_library or degree_schedule or moments_kernel_vs_reference" 2>&1 | tail -3
for hip in True False; do
python bench.py --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 10 --set graph.SORTED_COPY_IN_HIP=$hip 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('HIP=$hip', 'ms', round(d['ms_per_step'],4), d['amortised_setup_ms'], d['checksum'])"
done
python bench.py --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 10 --order sum_first 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sum_first ms', round(d['ms_per_step'],4), d['amortised_setup_ms'], d['checksum'])"
