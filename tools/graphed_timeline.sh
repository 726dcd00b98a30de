# Kernel timeline of one replayed training epoch of tools/graphed_step.py (start offsets, durations and gaps in us):
#   bash tools/graphed_timeline.sh OUT arxiv|cora|rmat|arxiv40
OUT=$1; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $OUT
rocprofv3 --kernel-trace -f csv -d $OUT -o tl -- python3 tools/graphed_step.py "$@" > $OUT/tl.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for fn in glob.glob("$OUT/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last replayed training step: between the last two optimizer updates (one fused launch each, graphed.FlatAdamStep)
adam = [i for i, r in enumerate(rows) if "FusedOptimizer" in r[2]]
last, first = adam[-1], adam[-2] + 1
t0 = rows[first][0]
prev_end = None
busy = 0
for s, e, k in rows[first:last + 1]:
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:7.1f}"
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  {gap:12s} {name}")
    prev_end = e
    busy += e - s
print(f"span {(rows[last][1] - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {last + 1 - first} kernels")
PY
rm -f $OUT/*kernel_trace.csv
