"""EXPERIMENT: the moment kernels when most nodes fall into the SAME piece of a feature (sparse / binary features:
bag-of-words, one-hot) — every lane of a wavefront then adds to the same LDS bin.
    python tools/experiments/moments_skew.py [nodes]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import gnan_amd  # noqa: E402,F401
from gnan_amd import functional, pwl  # noqa: E402
from gnan_amd.functional import StackedMLP  # noqa: E402

DEV = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
F, H, L = 64, 64, 3


def tables(C):
    gen = torch.Generator().manual_seed(0)
    w1 = torch.randn(F, H, generator=gen) * 1.4
    w2 = torch.randn(1, F, H, H, generator=gen) * (2.0 / (2 * H)) ** 0.5
    w3 = torch.randn(F, C, H, generator=gen) * (2.0 / (H + C)) ** 0.5
    b1, b2, b3 = torch.randn(F, H, generator=gen) * 0.5, torch.randn(1, F, H, generator=gen) * 0.5, torch.randn(F, C, generator=gen) * 0.5
    st = StackedMLP(w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), w3.to(DEV), b3.to(DEV), L, H, C, F)
    return pwl.build_tables(st)


def main():
    out = {"nodes": N}
    for C, n in ((1, N), (8, N // 8)):
        t = tables(C)
        g = torch.randn(n, C, device=DEV)
        for tag, x in (("uniform", torch.rand(n, F, device=DEV)),
                       ("95% zeros", torch.rand(n, F, device=DEV) * (torch.rand(n, F, device=DEV) < 0.05)),
                       ("constant", torch.full((n, F), 0.25, device=DEV))):
            xm = x.abs().max().double()
            for _ in range(2):
                functional._fpwl_moments(x, t, g, True, x_abs_max=xm, raw=True)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(5):
                functional._fpwl_moments(x, t, g, True, x_abs_max=xm, raw=True)
            ev[1].record()
            torch.cuda.synchronize()
            out[f"C={C} n={n} {tag}"] = round(ev[0].elapsed_time(ev[1]) / 5, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
