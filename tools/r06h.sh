OUT=gpurun_out/r06h; mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
run() { tag=$1; shift
env "$@" rocprofv3 --kernel-trace --stats -f csv -d $OUT/$tag -o pb -- python3 tools/pb_bench.py 1 > $OUT/$tag.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('gpurun_out/r06h/$tag/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pb_reduce' in r['Name']:
            print('$tag', r['Name'][:60], r['Calls'], r['AverageNs'])
PY
}
run base PB_X=1
run nocnt PB_USE_CNT=0
run norest PB_WITH_REST=0
run neither PB_USE_CNT=0 PB_WITH_REST=0
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
