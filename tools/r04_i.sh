OUT=gpurun_out/r04i
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python -m pytest tests/test_gpu_harness.py -x -q -m gpu > $OUT/test_harness.log 2>&1; echo "harness tests rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $OUT/test_multirank.log 2>&1; echo "multirank tests rc=$?" >> $OUT/rc.log
rocprofv3 --kernel-trace -f csv -d $OUT/trace8 -o t8 -- python3 bench.py --traffic committed --sustain-seconds 0 --steps 6 --warmup 2 --no-cpu-baseline --emulate-world 8 --partition halo > $OUT/trace8.log 2>&1
python3 - <<'PY' > $OUT/trace8_timeline.txt 2>&1
import csv, glob
fn = glob.glob("gpurun_out/r04i/trace8/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  stream {r.get('Stream_Id', '?'):>4}  {name}")
    prev_end = max(prev_end, e)
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/rc.log; tail -5 $OUT/test_harness.log; tail -30 $OUT/test_multirank.log | cut -c1-400; cat $OUT/trace8_timeline.txt
