OUT=gpurun_out/r06i; mkdir -p $OUT
python -m pytest tests -q -m gpu -x --durations=12 > $OUT/gpu_suite.log 2>&1; tail -22 $OUT/gpu_suite.log
