# Kernel timeline of one steady-state bench step (start offsets and durations in us):  bash tools/step_timeline.sh OUT [bench flags]
OUT=$1; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $OUT
rocprofv3 --kernel-trace -f csv -d $OUT -o tl -- python3 bench.py --traffic committed --sustain-seconds 0 --steps 6 --warmup 3 --no-cpu-baseline "$@" > $OUT/tl.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for fn in glob.glob("$OUT/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "pwl_build_kernel" in r[2]]
i0, i1 = idx[-2], idx[-1]                     # the last complete step
t0 = rows[i0][0]
prev_end = None
for s, e, k in rows[i0:i1]:
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:7.1f}"
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  {gap:12s} {name}")
    prev_end = e
PY
rm -f $OUT/*kernel_trace.csv
