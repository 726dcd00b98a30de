# L2 hit / miss and fabric request counters of the narrow (sum-first) aggregation: propagation-blocked kernels (default) and the
# row-parallel spmm_hot_kernel (aggregate.PB_NARROW=False), forward and training step.   bash tools/pmc_narrow.sh OUTDIR
OUT=$1
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_pb_$T -o p -- python3 bench.py --traffic off --sustain-seconds 0 --order sum_first --no-cpu-baseline --steps 3 --warmup 2 > $OUT/pmc_pb_$T.log 2>&1
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_rows_$T -o p -- python3 bench.py --traffic off --sustain-seconds 0 --order sum_first --no-cpu-baseline --steps 3 --warmup 2 --set aggregate.PB_NARROW=False > $OUT/pmc_rows_$T.log 2>&1
done
python3 - <<PY > $OUT/pmc_narrow.csv
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    leg = "propagation_blocked" if "/pmc_pb_" in fn else "row_parallel"
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "spmm" in k or "pb_" in k:
            acc[(leg, k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("leg,kernel,counter,mean_per_launch,launches")
for (leg, k, c), v in sorted(acc.items()):
    print(f"{leg},\"{k}\",{c},{sum(v)/len(v):.6g},{len(v)}")
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
