OUT=gpurun_out/r06f; mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python -m pytest tests/test_gpu_kernels.py -q -x -k "propagation_blocked" 2>&1 | tail -3
PB_FLAG_SETS=0,1,2,3,4,0x200,0x800 python tools/pb_bench.py 1 2>&1 | tail -1
PB_FLAG_SETS=0,3 python tools/pb_bench.py 2 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/pb -o pb -- python3 tools/pb_bench.py 1 > $OUT/pb_prof.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r06f/pb/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pb_' in r['Name'] or 'spmm' in r['Name']:
            print(r['Name'][:90], r['Calls'], r['AverageNs'])
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
