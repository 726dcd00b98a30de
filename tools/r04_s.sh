#!/bin/bash
# moments kernel with the next round's loads requested early: kernel stats of a C4 training step + the table-path tests
OUT=gpurun_out/r04s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -f csv -d $OUT/train -o tr -- python3 tools/train_step_c4.py > $OUT/train_step.log 2>&1
python3 - <<'PY' > $OUT/train_kernels.txt
import csv, glob
for f in glob.glob('gpurun_out/r04s/train/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
python3 tools/train_step_c4.py > $OUT/train_step_noprof.log 2>&1
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "moment or grad or backward or table" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
cat $OUT/train_kernels.txt; tail -2 $OUT/train_step_noprof.log
