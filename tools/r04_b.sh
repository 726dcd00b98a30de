OUT=gpurun_out/r04b
mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "direct_index or feature_range" > $OUT/test_index.log 2>&1; echo "index tests rc=$?" >> $OUT/rc.log
python tools/lookup_ab.py > $OUT/lookup_ab.jsonl 2> $OUT/lookup_ab.err; echo "ab rc=$?" >> $OUT/rc.log
python bench.py --traffic committed --sustain-seconds 0 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.log 2>&1; echo "bench c4 rc=$?" >> $OUT/rc.log
python bench.py --config c2 --steps 2 --warmup 1 --graphs 600 > $OUT/bench_c2.log 2>&1; echo "bench c2 rc=$?" >> $OUT/rc.log
cat $OUT/rc.log; tail -5 $OUT/test_index.log; cat $OUT/lookup_ab.jsonl
