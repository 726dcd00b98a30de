# round 6, first GPU contact: (1) kernel timeline of the slowest 8-rank share (replayed, RCCL group on), (2) L2 hit / miss /
# fetch counters of the narrow aggregation kernels (sum-first forward and training step), (3) coverage of the hottest columns.
OUT=gpurun_out/r06a; mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
RANK=2 bash tools/step_timeline.sh $OUT/share_tl --emulate-world 8 --partition halo --force-dist > $OUT/share_timeline.txt 2>&1
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_sf_$T -o p -- python3 bench.py --traffic off --sustain-seconds 0 --order sum_first --no-cpu-baseline --steps 3 --warmup 2 > $OUT/pmc_sf_$T.log 2>&1
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_tr_$T -o p -- python3 tools/train_step_c4.py > $OUT/pmc_tr_$T.log 2>&1
done
python3 - <<'PY' > gpurun_out/r06a/pmc_spmm_hot.csv
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("gpurun_out/r06a/pmc_*/**/*counter_collection.csv", recursive=True):
    leg = "sum_first_fwd" if "/pmc_sf_" in fn else "train_step"
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "spmm" in k or "fpwl_index" in k or "fpwl_moments" in k:
            acc[(leg, k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("leg,kernel,counter,mean_per_launch,launches")
for (leg, k, c), v in sorted(acc.items()):
    print(f"{leg},\"{k}\",{c},{sum(v)/len(v):.6g},{len(v)}")
PY
python3 tools/hot_coverage.py > $OUT/hot_coverage.json 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
