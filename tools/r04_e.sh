OUT=gpurun_out/r04e
mkdir -p $OUT
tools/_bin/lookup_ceiling > $OUT/ceiling.jsonl 2>&1
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "direct_index or feature_range" > $OUT/test_index.log 2>&1; echo "index tests rc=$?" >> $OUT/rc.log
python tools/debug_c2.py > $OUT/debug_c2.log 2>&1
cat $OUT/rc.log; tail -5 $OUT/test_index.log; grep "<<<<" $OUT/debug_c2.log | head -20; tail -3 $OUT/debug_c2.log
