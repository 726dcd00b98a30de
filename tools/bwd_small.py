"""Small-batch parameter gradients: gnan_fmlp_bwd vs the batched-GEMM restatement differentiated by torch (device time)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnan_amd  # noqa: F401
from gnan_amd import functional
from gnan_amd.functional import StackedMLP, feature_mlps

def stack(F, H, C, dev):
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g, device=dev) * 0.3
    return StackedMLP(r(F, H), r(F, H), r(1, F, H, H), r(1, F, H), r(F, C, H), r(F, C), 3, H, C, F)

dev = "cuda"
for n, F, C in ((30, 15, 1), (3000, 129, 1), (2708, 1434, 7), (16000, 64, 4)):
    x = torch.rand(n, F, device=dev)
    res = {"n": n, "F": F, "C": C}
    for tag, on in (("hip_ms", True), ("torch_ms", False)):
        functional.HIP_SMALL_BACKWARD = on
        functional.FMLP_ALGO = 0
        functional.PWL_MIN_WORK_GRAD = 1 << 40            # keep the direct forward for every size here
        st = stack(F, 64, C, dev)
        leaves = [t.requires_grad_(True) for t in st[:6]]
        gup = torch.randn(n, C, device=dev)
        ts = []
        for rep in range(6):
            out = feature_mlps(x, st, True)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            torch.autograd.grad(out, leaves, gup)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[tag] = round(sorted(ts)[1], 3)
    print(json.dumps(res))
