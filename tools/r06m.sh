python -m pytest tests/test_gpu_multirank.py -q -x -k "bench_starts" 2>&1 | tail -3
python bench.py --force-dist --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 20 --partition halo 2>&1 | grep -E "^\{|so far" | cut -c1-200 | tail -3
python bench.py --force-dist --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 20 --partition halo 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(json.dumps(d['alt_partitions'])[:1500]); print(d['ms_per_step'], d['cut'], d['owned_rows_rank0'])"
