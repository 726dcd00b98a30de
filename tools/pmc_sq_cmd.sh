# SQ-side counters for the kernels of any command:  bash tools/pmc_sq_cmd.sh OUTDIR python3 script.py [args]
OUT=$1; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -f csv -d $OUT -o sq -- "$@" > $OUT/sq.log 2>&1
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
ours = ("spmm", "fpwl", "pwl_", "colsum", "fmlp", "dense_to_code", "bfs_")
for fn in glob.glob("$OUT/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if any(o in k for o in ours):
            acc[(k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
names = sorted({k for k, _ in acc})
cs = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VALU"]
print("kernel".ljust(62), " ".join(c[3:].rjust(18) for c in cs))
for k in names:
    print(k.ljust(62), " ".join(f"{sum(acc[(k, c)])/max(len(acc[(k, c)]),1):18.4g}" for c in cs))
PY
rm -f $OUT/*kernel_trace.csv
