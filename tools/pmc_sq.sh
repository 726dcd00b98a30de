# SQ-side counters of the default bench's kernels (one pass, 8 counters): where do the waves of a kernel spend their cycles?
OUT=${1:-gpurun_out/sq}
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -f csv -d $OUT -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq.log 2>&1
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for fn in glob.glob("$OUT/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if any(o in k for o in ("spmm_kernel", "fpwl_fast", "pwl_build")):
            acc[(k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(k, c, f"{sum(v)/len(v):.4g}")
PY
rm -f $OUT/*kernel_trace.csv
