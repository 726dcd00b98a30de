mkdir -p gpurun_out/r04d
python tools/debug_c2.py > gpurun_out/r04d/debug_c2.log 2>&1
tail -70 gpurun_out/r04d/debug_c2.log
