# the driver's own commands, once, before the final pass
mkdir -p gpurun_out/r05p
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r05p/bench_full.json 2> gpurun_out/r05p/bench_full.err; tail -3 gpurun_out/r05p/bench_full.err | cut -c1-300; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r05p/bench_full.json') if l.startswith('{')][-1]); print({k: d.get(k) for k in ('value','ms_per_step','stages_ms','sustained_ms_per_step','amortised_setup_ms')}); print(d['roofline']); print({k: d['cpu_baseline'][k] for k in ('value','cores','dominant_leg','parity_max_rel_err')})"
timeout 600 python bench.py --config c3 --loop reference > gpurun_out/r05p/c3.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05p/c3.json')); print({k: d[k] for k in ('value','fwd_ms','fwd_bwd_ms','ms_per_step','replayed_fwd_bwd_ms','replay_note')}); print(d['roofline']); print(d['cpu_baseline']['parity_max_rel_err'])"
timeout 900 python bench.py --config c5 --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c5', d['ms_per_step'], d['stages_ms'])"
