"""Host profile (cProfile, own time) of bench.py's default run: where the HOST spends a step when a stage's device time is
short but its wall time is not.    python tools/bench_host_profile.py [bench flags]"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--traffic", "committed", "--sustain-seconds", "0"] + sys.argv[1:]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats("tottime")
    st.print_stats(22)
    st.print_callers("torch.empty")
    import torch
    ms = torch.cuda.memory_stats()
    sys.stderr.write("allocator: " + " ".join(f"{k}={ms.get(k)}" for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms",
                                                                          "reserved_bytes.all.peak", "reserved_bytes.all.current",
                                                                          "allocated_bytes.all.peak")) + "\n")
    sys.stderr.write(s.getvalue())
