#!/bin/bash
# kernel timeline of one replayed forward + backward step of config 3 (bench.py --config c3)
OUT=gpurun_out/c3tl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace -f csv -d $OUT/t -o k -- python3 bench.py --config c3 --no-cpu-baseline --steps 5 --warmup 2 "$@" > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/c3tl/t/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    import re
    note = open('gpurun_out/c3tl/log.txt').read()
    m = re.search(r'(\d+) kernels per replay', note)
    n = int(m.group(1)) if m else 36
    last = rows[-(n + 4):]            # (the four launches after the last replay: the bench's own checksum / copies)
    t0 = int(last[0]['Start_Timestamp'])
    prev_end = t0
    for r in last:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name[:90]}")
        prev_end = e
    print('total', (int(last[-1]['End_Timestamp']) - t0) / 1e3)
PY
rm -rf $OUT/t
