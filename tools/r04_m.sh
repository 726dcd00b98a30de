OUT=gpurun_out/r04m
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batched" > $OUT/test_batched.log 2>&1; echo "batched tests rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_multirank.py -q -m gpu > $OUT/test_multirank.log 2>&1; echo "multirank tests rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "rest_bucket or direct_index" > $OUT/test_rest.log 2>&1; echo "rest/index tests rc=$?" >> $OUT/rc.log
python tools/batched_bench.py > $OUT/batched_bench.jsonl 2> $OUT/batched_bench.err
python tools/lookup_ab.py > $OUT/lookup_ab.jsonl 2> $OUT/lookup_ab.err
bash tools/emulate_shares.sh --share-fork fmlp > $OUT/emulated_shares.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off --sustain-seconds 0 > $OUT/bench_c4.log 2>&1
cat $OUT/rc.log; tail -3 $OUT/test_batched.log; tail -6 $OUT/test_multirank.log | cut -c1-250; tail -3 $OUT/test_rest.log; cat $OUT/batched_bench.jsonl; cat $OUT/lookup_ab.jsonl | cut -c1-400; cat $OUT/emulated_shares.txt
grep "^{" $OUT/bench_c4.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'], d['roofline']['frac'])"
