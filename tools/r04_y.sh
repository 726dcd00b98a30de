#!/bin/bash
OUT=gpurun_out/r04y; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "small_graph or small_golden or nam_readout" > $OUT/t.log 2>&1; tail -2 $OUT/t.log
bash tools/r04_x.sh
for v in standalone nam; do python tools/muta_epoch.py 4337 $v > $OUT/muta_$v.json 2> $OUT/err_$v.log; python - $v <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r04y/muta_{sys.argv[1]}.json').read())
print(sys.argv[1], d['train_ms_per_graph_by_epoch'], d['eval_ms_per_graph_by_epoch'], d['kernels_per_step_histogram'])
PY
done
