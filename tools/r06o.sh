python -m pytest tests/test_gpu_kernels.py -q -x -k "by_the_library" 2>&1 | tail -3
python - <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
import gnan_amd
from gnan_amd import synthetic as syn, graph as G
src, dst = syn.rmat_edges(24, 10_000_000, 100_000_000, seed=0, device="cuda")
for hip in (True, False, True):
    G.TRANSPOSE_IN_HIP = hip
    g = syn.hop1_csr(src, dst, 10_000_000)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = g.transposed()
    torch.cuda.synchronize(); print("transposed HIP" if hip else "transposed torch", round((time.perf_counter() - t0) * 1e3, 1), "ms")
    t0 = time.perf_counter(); p = t.pb_plan(2); torch.cuda.synchronize(); print("  pb_plan(2) of the transpose", round((time.perf_counter() - t0) * 1e3, 1), "ms")
    t0 = time.perf_counter(); p = g.pb_plan(1); torch.cuda.synchronize(); print("  pb_plan(1)", round((time.perf_counter() - t0) * 1e3, 1), "ms")
    t0 = time.perf_counter(); p = g.degree_sorted_copy(); torch.cuda.synchronize(); print("  degree_sorted_copy", round((time.perf_counter() - t0) * 1e3, 1), "ms")
PY
