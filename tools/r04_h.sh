OUT=gpurun_out/r04h
mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "direct_index or feature_range" > $OUT/test_index.log 2>&1; echo "index tests rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_harness.py -x -q -m gpu -k "run_exp" > $OUT/test_run.log 2>&1; echo "run_exp tests rc=$?" >> $OUT/rc.log
bash tools/emulate_shares.sh > $OUT/emulated_shares.txt 2>&1
SHARE_GRAPH=off bash tools/emulate_shares.sh > $OUT/emulated_shares_eager.txt 2>&1
python bench.py --traffic committed --sustain-seconds 0 --steps 5 --warmup 2 --no-cpu-baseline --emulate-world 8 --partition halo > $OUT/emu8.log 2>&1
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $OUT/test_multirank.log 2>&1; echo "multirank tests rc=$?" >> $OUT/rc.log
cat $OUT/rc.log; tail -5 $OUT/test_index.log; tail -15 $OUT/test_run.log; cat $OUT/emulated_shares.txt $OUT/emulated_shares_eager.txt; tail -5 $OUT/emu8.log | cut -c1-600; tail -5 $OUT/test_multirank.log
