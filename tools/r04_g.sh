OUT=gpurun_out/r04g
mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "direct_index or feature_range" > $OUT/test_index.log 2>&1; echo "index tests rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_harness.py -x -q -m gpu -k "run_exp" > $OUT/test_run.log 2>&1; echo "run_exp tests rc=$?" >> $OUT/rc.log
python tools/lookup_ab.py > $OUT/lookup_ab.jsonl 2> $OUT/lookup_ab.err; echo "ab rc=$?" >> $OUT/rc.log
python bench.py --traffic committed --sustain-seconds 0 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.log 2>&1; echo "bench c4 rc=$?" >> $OUT/rc.log
python bench.py --traffic committed --sustain-seconds 0 --steps 10 --warmup 3 --no-cpu-baseline --order sum_first > $OUT/bench_c4_sum.log 2>&1; echo "bench c4 sum rc=$?" >> $OUT/rc.log
python tools/train_step_c4.py > $OUT/train_step.log 2>&1; echo "train step rc=$?" >> $OUT/rc.log
bash tools/emulate_shares.sh > $OUT/emulated_shares.txt 2>&1
cat $OUT/rc.log; tail -5 $OUT/test_index.log; tail -15 $OUT/test_run.log; cat $OUT/lookup_ab.jsonl; cat $OUT/emulated_shares.txt; tail -3 $OUT/train_step.log
