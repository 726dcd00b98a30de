#!/bin/bash
# per-kernel time and SQ counters of the C4 training step (tools/train_step_c4.py) on the working tree
OUT=gpurun_out/tk; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -f csv -d $OUT/tr -o tr -- python3 tools/train_step_c4.py > $OUT/log.txt 2>&1
f=$(find $OUT/tr -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $OUT/kernels.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if any(k in n for k in ("fpwl", "spmm", "pack_bwd", "colsum", "pwl_", "absmax", "scales")):
        print(f"{n[:72]:72s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs']) / 1e3:9.1f}")
PY
cat $OUT/kernels.txt
bash tools/pmc_sq_cmd.sh $OUT/sq python3 tools/train_step_c4.py > $OUT/sq.txt 2>&1
grep -E "kernel|fpwl_moments|fpwl_index" $OUT/sq.txt
rm -rf $OUT/tr
