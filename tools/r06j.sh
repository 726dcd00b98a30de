OUT=gpurun_out/r06j; mkdir -p $OUT
python -m pytest tests -q -m gpu --durations=5 > $OUT/gpu_suite.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/gpu_suite.log | tail -40
