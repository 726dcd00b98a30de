#!/usr/bin/env python3
"""Forward+backward of models.TensorGNAN on the C4-shaped graph (R-MAT 10M / 100M, F = 64): where a training step's time goes.
    python tools/train_step_c4.py [nodes edges scale]        (under rocprofv3 --kernel-trace --stats for the kernel split)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gnan_amd  # noqa: E402,F401
from gnan_amd import synthetic as syn  # noqa: E402
from gnan_amd.models import TensorGNAN  # noqa: E402

DEV = "cuda"
if "PB_OFF" in os.environ:                 # A/B: the row-parallel narrow aggregation instead of csrc/spmm_pb.hip
    from gnan_amd import aggregate as _agg
    _agg.PB_NARROW = False
N, E, SCALE = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (10_000_000, 100_000_000, 24)))
F = 64


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def main():
    torch.manual_seed(0)
    src, dst = syn.rmat_edges(SCALE, N, E, seed=0, device=DEV)
    g = syn.hop1_csr(src, dst, N)
    del src, dst
    x = syn.block_features(N, F, 0, N, seed=1, device=DEV)
    y = torch.randn(N, 1, device=DEV)
    m = TensorGNAN(F, 1, 3, hidden_channels=64, device=DEV)
    with torch.no_grad():
        for _, p in m.named_parameters():
            torch.nn.init.xavier_normal_(p, gain=1.0) if p.dim() == 2 else p.normal_(0.0, 0.5)
    m = m.to(DEV).eval()                       # eval: the reference trains without Dropout after its first epoch
    # the optimizer over the flat parameter buffers (gnan_amd.optim_params: a dozen tensors) — what DESIGN.md quotes, the kernels'
    # step; PER_LAYER=1: over model.parameters() as main.py:141 writes it (torch's per-tensor bookkeeping on ~390 tensors on top)
    per_layer = bool(os.environ.get("PER_LAYER"))
    opt = torch.optim.Adam(m.parameters() if per_layer else gnan_amd.optim_params(m), lr=1e-3)
    d = Bag(x=x, edge_index=None, gnan_graph=g)
    out = {"what": f"train_step_rmat_{N}n_{E}e_F{F}", "optimizer_over": "model.parameters()" if per_layer else "gnan_amd.optim_params(model)",
           "optimizer_tensors": sum(len(gr["params"]) for gr in opt.param_groups)}

    def fwd():
        with torch.no_grad():
            return m.forward(d)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m.forward(d), y)
        loss.backward()
        opt.step()
        return loss

    for name, fn, reps in (("fwd_ms", fwd, 20), ("fwd_bwd_adam_ms", step, 20)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for r in range(reps):
            fn()
            marks[r + 1].record()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / reps * 1e3                          # wall clock, mean
        per = sorted(marks[r].elapsed_time(marks[r + 1]) for r in range(reps))
        out[name.replace("_ms", "_device_ms")] = {"min": round(per[0], 3), "median": round(per[reps // 2], 3)}
    out["loss"] = float(step())
    print(json.dumps(out))
    if os.environ.get("GNAN_STEP_PROFILE"):        # kernel split of the steady-state step (torch.profiler, 3 steps)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:28]
        for e in rows:
            print(f"{e.device_time_total / 3e3:9.3f} ms/step  x{e.count / 3:6.1f}  {e.key[:110]}")


if __name__ == "__main__":
    main()
