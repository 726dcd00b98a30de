#!/bin/bash
OUT=gpurun_out/r04bk; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace -f csv -d $OUT/t -o k -- python3 tools/batch_kernel_probe.py $1 > $OUT/log.txt 2>&1
grep CASE $OUT/log.txt
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r04bk/t/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    fw = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'small_graph_batch_kernel' in r['Kernel_Name']]
    bw = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'small_graph_batch_bwd_kernel' in r['Kernel_Name']]
    print('fwd', [round(min(fw[i:i+4]),1) for i in range(0, len(fw), 4)])
    print('bwd', [round(min(bw[i:i+4]),1) for i in range(0, len(bw), 4)])
PY
rm -rf $OUT/t
