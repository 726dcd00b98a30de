#!/bin/bash
# after the module split: every bench configuration once, the emulated 8-rank share, the launcher with 2 ranks on one GPU
OUT=gpurun_out/r04u; mkdir -p $OUT
python bench.py --traffic off --sustain-seconds 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; tail -c 600 $OUT/bench_c4.json; echo
python bench.py --config c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 300 $OUT/bench_c3.json; echo
python bench.py --config c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 300 $OUT/bench_c2.json; echo
python bench.py --emulate-world 8 --no-cpu-baseline --traffic off --steps 50 --warmup 10 > $OUT/emu8.json 2> $OUT/emu8.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04u/emu8.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('ms_per_step','value','share_replayed_from_hipgraphs','share_graph_note','share_guard_tripped')})
PY
python bench.py --gpus 2 --same-device --backend gloo --no-cpu-baseline --traffic off --steps 5 --warmup 2 --nodes 200000 --edges 2000000 --scale 18 > $OUT/two.json 2> $OUT/two.err; tail -c 400 $OUT/two.json; echo; tail -3 $OUT/two.err
