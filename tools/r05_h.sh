# round 5, eighth GPU call: two-tier graph slots + direct gradient write, label flag; then the whole suite with durations
mkdir -p gpurun_out/r05h
timeout 1800 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_harness.py -q -m gpu -x 2>&1 | grep -v "Warning\|warn" | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batched" 2>&1 | tail -4
timeout 900 python tools/muta_epoch.py > gpurun_out/r05h/muta_epoch.json 2>/dev/null; cat gpurun_out/r05h/muta_epoch.json | cut -c1-1500
timeout 900 python bench.py --config c2 --no-cpu-baseline > gpurun_out/r05h/c2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05h/c2.json')); print({k: d[k] for k in ('ms_per_graph','replayed_eval_ms_per_graph','replayed_train_ms_per_graph','kernels_per_replayed_training_step','captured_training_steps')})"
GNAN_SLOT=0 timeout 900 python - <<'PY' 2>/dev/null
import sys, os, json, subprocess
sys.path.insert(0, 'tools')
import gnan_amd
from gnan_amd import harness
harness.SLOT_STEPS = False
sys.argv = ['muta_epoch.py']
import runpy
runpy.run_path('tools/muta_epoch.py', run_name='__main__')
PY
timeout 2400 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r05h/gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r05h/gpu_suite.txt; grep -v "Warning\|warn" gpurun_out/r05h/gpu_suite.txt | tail -30
