#!/bin/bash
OUT=gpurun_out/r04bp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
cd tools && rocprofv3 --kernel-trace -f csv -d ../$OUT/t -o k -- python3 batched_replay_profile.py 32 > ../$OUT/log.txt 2>&1; cd ..
grep "kernels per step" $OUT/log.txt
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r04bp/t/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    last = rows[-12:]
    t0 = int(last[0]['Start_Timestamp']); prev = t0
    for r in last:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {name[:100]}")
        prev = e
PY
rm -rf $OUT/t
