OUT=gpurun_out/r04k
mkdir -p $OUT
python tools/debug_refacc.py > $OUT/debug_refacc.log 2>&1
python bench.py --traffic committed --sustain-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c4.log 2>&1; echo "bench c4 rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spmm or fused or hub or aggregation or readout or bf16 or packed or degree" > $OUT/test_spmm.log 2>&1; echo "spmm tests rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/test_parity.log 2>&1; echo "parity tests rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_multirank.py -q -m gpu > $OUT/test_multirank.log 2>&1; echo "multirank tests rc=$?" >> $OUT/rc.log
python -m pytest tests/test_gpu_harness.py -q -m gpu > $OUT/test_harness.log 2>&1; echo "harness tests rc=$?" >> $OUT/rc.log
python tools/train_step_c4_reference_order.py > $OUT/train_ref.log 2>&1
cat $OUT/rc.log; tail -9 $OUT/debug_refacc.log; grep "^{" $OUT/bench_c4.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'], d['roofline']['frac'])"; tail -4 $OUT/test_spmm.log; tail -4 $OUT/test_parity.log; tail -6 $OUT/test_multirank.log | cut -c1-200; tail -4 $OUT/test_harness.log; tail -2 $OUT/train_ref.log
( time python bench.py ) > gpurun_out/r04k/bench_default.log 2>&1
tail -5 gpurun_out/r04k/bench_default.log | cut -c1-3000
