#!/bin/bash
OUT=gpurun_out/r04x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace -f csv -d $OUT/t -o k -- python3 tools/pre_rho_bwd_time.py 30 > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r04x/t/**/*kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'small_graph' in r['Kernel_Name']]
    # three phases of 30 iterations: fwd, bwd alternating
    seq = [(r['Kernel_Name'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Grid_Size_X') or r.get('Grid_Size')) for r in rows]
    for ph in range(3):
        part = seq[ph*60:(ph+1)*60]
        fw = [d for n,d,g in part if 'bwd' not in n][5:]; bw = [d for n,d,g in part if 'bwd' in n][5:]
        grid = [g for n,d,g in part if 'bwd' in n][:1]
        print(ph, 'fwd_us', round(min(fw),1), 'bwd_us', round(min(bw),1), round(sorted(bw)[len(bw)//2],1), 'bwd grid', grid)
PY
rm -rf $OUT/t
