OUT=gpurun_out/r04c
mkdir -p $OUT
tools/_bin/lookup_ceiling > $OUT/ceiling.jsonl 2>&1
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "direct_index or feature_range" > $OUT/test_index.log 2>&1; echo "index tests rc=$?" >> $OUT/rc.log
cat $OUT/rc.log; tail -5 $OUT/test_index.log; cat $OUT/ceiling.jsonl
