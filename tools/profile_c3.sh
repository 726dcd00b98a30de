# rocprofv3 evidence for config 3 (arxiv-shaped forward+backward, C = 1): kernel stats + HBM bytes per kernel.
#   bash tools/profile_c3.sh gpurun_out/r02c_c3      (on the GPU box; three passes)
OUT=${1:-gpurun_out/c3}
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- python3 tools/profile_arxiv.py > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o fetch -- python3 tools/profile_arxiv.py > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o write -- python3 tools/profile_arxiv.py > $OUT/write.log 2>&1
python3 - <<PY
import csv, glob, collections
ours = ("spmm", "fpwl", "pwl_", "colsum", "absmax", "pack_bwd", "scales")
dur = {}
for r in csv.DictReader(open(glob.glob("$OUT/stats/*_kernel_stats.csv")[0])):
    if any(o in r["Name"] for o in ours):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
acc = collections.defaultdict(list)
for d in ("fetch", "write"):
    for fn in glob.glob("$OUT/%s/*_counter_collection.csv" % d):
        for r in csv.DictReader(open(fn)):
            if any(o in r["Kernel_Name"] for o in ours):
                acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$OUT/c3_hbm_summary.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "avg_us", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "read_MB_lo_hi", "write_MB", "GBps_lo_hi"])
    for k, (calls, ns) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        fe = acc.get((k, "FETCH_SIZE"), [0]); wr = acc.get((k, "WRITE_SIZE"), [0])
        fe, wr = sum(fe) / len(fe), sum(wr) / len(wr)
        lo, hi = (fe + wr) * 1024 / ns, (2 * fe + wr) * 1024 / ns        # bytes per ns = GB/s; gfx950: FETCH_SIZE counts 64 B per 128-B request at most
        w.writerow([k.replace("(anonymous namespace)::", "")[:90], calls, round(ns / 1e3, 1), round(fe, 1), round(wr, 1),
                    f"{fe * 1024 / 1e6:.1f}-{2 * fe * 1024 / 1e6:.1f}", f"{wr * 1024 / 1e6:.1f}", f"{lo:.0f}-{hi:.0f}"])
print(open("$OUT/c3_hbm_summary.csv").read())
PY
find $OUT -name "*_kernel_trace.csv" -delete
