#!/bin/bash
# kernel durations of the replayed Mutagenicity-shaped steps, per model variant
OUT=gpurun_out/r04w; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in models standalone nam; do
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/$v -o k -- python3 tools/muta_epoch.py 600 $v > $OUT/$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
for f in glob.glob(f'gpurun_out/r04w/{v}/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(v)
    for r in rows[:8]:
        print(f"   {r['Name'][:80]:80s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f} pct {r['Percentage']}")
PY
done
