OUT=gpurun_out/r04l
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batched" > $OUT/test_batched.log 2>&1; echo "batched tests rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "small_graph" > $OUT/test_small.log 2>&1; echo "small graph tests rc=$?" >> $OUT/rc.log
python tools/batched_bench.py > $OUT/batched_bench.jsonl 2> $OUT/batched_bench.err; echo "batched bench rc=$?" >> $OUT/rc.log
python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_configs.py::test_c5_papers100m_shaped_bf16 --deselect tests/test_gpu_configs.py::test_more_than_2_31_listed_pairs > $OUT/test_all.log 2>&1; echo "all gpu tests (minus the two giant ones) rc=$?" >> $OUT/rc.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off --sustain-seconds 0 > $OUT/bench_c4.log 2>&1
cat $OUT/rc.log; tail -5 $OUT/test_batched.log; tail -5 $OUT/test_small.log; cat $OUT/batched_bench.jsonl; tail -3 $OUT/batched_bench.err; tail -8 $OUT/test_all.log | cut -c1-300
grep "^{" $OUT/bench_c4.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'], d['roofline']['frac'])"
