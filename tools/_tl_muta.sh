OUT=gpurun_out/r03y/tl_muta
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $OUT
rocprofv3 --kernel-trace -f csv -d $OUT -o tl -- python3 tools/graphed_step.py muta > $OUT/tl.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for fn in glob.glob("$OUT/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "FusedOptimizer" in r[2]]
last, first = adam[-1], adam[-2] + 1
t0 = rows[first][0]; prev=None
for s, e, k in rows[first:last + 1]:
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
    gap = "" if prev is None else f"gap {(s - prev) / 1e3:6.1f}"
    print(f"{(s - t0) / 1e3:8.1f} us dur {(e - s) / 1e3:6.1f} {gap:11s} {name}")
    prev = e
PY
rm -f $OUT/*kernel_trace.csv
