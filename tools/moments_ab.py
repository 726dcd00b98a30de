#!/usr/bin/env python3
"""A/B of the table path's moment kernels on a C4-sized batch (10M nodes x 64 features, H = 64, L = 3):
    python tools/moments_ab.py [nodes]        functional.MOMENTS_GENERAL: the general kernel"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gnan_amd  # noqa: E402,F401
from gnan_amd import functional, pwl  # noqa: E402
from gnan_amd.functional import StackedMLP  # noqa: E402

DEV = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
F, H, L = 64, 64, 3


def main():
    gen = torch.Generator().manual_seed(0)
    w1 = torch.randn(F, H, generator=gen) * 1.4
    w2 = torch.randn(1, F, H, H, generator=gen) * (2.0 / (2 * H)) ** 0.5
    w3 = torch.randn(F, 1, H, generator=gen) * (2.0 / (H + 1)) ** 0.5
    b1, b2, b3 = torch.randn(F, H, generator=gen) * 0.5, torch.randn(1, F, H, generator=gen) * 0.5, torch.randn(F, 1, generator=gen) * 0.5
    st = StackedMLP(w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), w3.to(DEV), b3.to(DEV), L, H, 1, F)
    t = pwl.build_tables(st)
    x = torch.rand(N, F, device=DEV)
    xm = x.abs().max().double()
    out = {"nodes": N, "pieces": int(t.anchor.numel()), "max_pieces": int(t.max_pieces)}
    for sum_features in (True, False):
        g = torch.randn(N, 1 if sum_features else F, device=DEV)
        ref = None
        for bs in ("1", "0"):
            functional.MOMENTS_GENERAL = bs == "1"             # gnan_fpwl_args.flags & GNAN_FPWL_MOMENTS_GENERAL
            for _ in range(2):
                M = functional._fpwl_moments(x, t, g, sum_features, x_abs_max=xm, raw=True)[0]
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(5):
                M = functional._fpwl_moments(x, t, g, sum_features, x_abs_max=xm, raw=True)[0]
            ev[1].record()
            torch.cuda.synchronize()
            if ref is None:
                ref = M
            out[f"sum={int(sum_features)} general={bs}"] = {"ms": round(ev[0].elapsed_time(ev[1]) / 5, 4), "same_bits": bool(torch.equal(M, ref))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
