OUT=gpurun_out/r06k; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -q -x -k "moments_kernel_vs_reference" 2>&1 | tail -2
for opt in "" "--flat"; do
python tools/reference_loop_bench.py arxiv cora $opt 2>/dev/null | grep -v "by phase" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$opt', d['what'], 'persistent: anomaly', d['anomaly_ms_per_step'], 'plain', d['plain_ms_per_step'], 'opt tensors', d['optimizer_tensors'])"
python tools/reference_loop_bench.py arxiv cora muta --fresh-inputs $opt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$opt', d['what'], 'FRESH: upload', d['upload_ms_per_step'], 'anomaly', d['anomaly_ms_per_step'], 'plain', d['plain_ms_per_step'], 'plain w/o upload', d['plain_ms_per_step_without_upload'], 'by epoch', d['plain_ms_per_step_by_epoch'])"
done
