# round 4, first GPU call: new tests, bench --config lines, SQ counters of the look-up / moment kernels BEFORE the rewrite
OUT=gpurun_out/r04a
mkdir -p $OUT
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c3" > $OUT/test_c3.log 2>&1; echo "c3 test rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "bench_starts" > $OUT/test_launch.log 2>&1; echo "launcher test rc=$?" >> $OUT/rc.log
python bench.py --config c3 --steps 20 --warmup 5 > $OUT/bench_c3.log 2>&1; echo "bench c3 rc=$?" >> $OUT/rc.log
python bench.py --config c3 --out 40 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_c40.log 2>&1; echo "bench c3 c40 rc=$?" >> $OUT/rc.log
python bench.py --config c2 --steps 3 --warmup 1 > $OUT/bench_c2.log 2>&1; echo "bench c2 rc=$?" >> $OUT/rc.log
bash tools/pmc_sq_cmd.sh $OUT/sq_fwd python3 bench.py --traffic committed --sustain-seconds 0 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq_fwd.txt 2>&1
bash tools/pmc_sq_cmd.sh $OUT/sq_train python3 tools/train_step_c4.py > $OUT/sq_train.txt 2>&1
python bench.py --traffic committed --sustain-seconds 0 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.log 2>&1; echo "bench c4 rc=$?" >> $OUT/rc.log
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -delete
cat $OUT/rc.log
