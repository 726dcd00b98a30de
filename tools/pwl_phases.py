"""Device time of gnan_pwl_build (build + compact) for the arxiv-shaped and the C4-shaped shape functions."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import reference_loop_bench as rb
import gnan_amd
from gnan_amd import pwl
for F in (129, 128, 64):
    m = rb.model_for(F, 1, False)
    st = m._stacked("fs", m.fs)
    for _ in range(3):
        t = pwl.build_tables(st, use_graph=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        pend = pwl.build_tables_lazy(st)
    e1.record()
    torch.cuda.synchronize()
    off = t.off.cpu()
    d = off[1:] - off[:-1]
    print("F", F, "build+compact+readback us per call", round(e0.elapsed_time(e1) * 1e3 / 50, 1), "pieces per feature: min %d median %d max %d" % (int(d.min()), int(d.median()), int(d.max())))
