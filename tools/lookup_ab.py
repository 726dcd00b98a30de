#!/usr/bin/env python3
"""A/B of the shape-function look-up on the C4 shape (10M nodes x 64 features, H = 64, L = 3, one channel): tree search
(fpwl_fast_kernel) against the direct-index kernel (fpwl_index_kernel) for several grid sizes.  Device time per launch
(HIP events, median of 20), bytes = 4 B of x in + 4 B (rows) / 2 B (bf16 rows) / 4/F B (feature sum) out per look-up."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnan_amd  # noqa: E402,F401
from gnan_amd import functional, pwl  # noqa: E402
from gnan_amd import synthetic as syn  # noqa: E402
from gnan_amd.functional import stack_mlps  # noqa: E402
from gnan_amd.models import TensorGNAN  # noqa: E402

DEV = "cuda"
N = int(os.environ.get("AB_NODES", 10_000_000))
F = 64


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    torch.manual_seed(0)
    m = TensorGNAN(F, 1, 3, hidden_channels=64, device=DEV)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() == 2:
                torch.nn.init.xavier_normal_(p, gain=1.0)
            else:
                p.normal_(0.0, 0.5)
    m = m.to(DEV).eval()
    x = syn.block_features(N, F, 0, N, 1, DEV)
    with torch.no_grad():
        st = stack_mlps(m.fs)
        tb = pwl.build_tables(st)
    rng = functional._feature_range(x)
    modes = {"rows": (False, torch.float32, 8.0), "rows_bf16": (False, torch.bfloat16, 6.0), "sum": (True, torch.float32, 4.0 + 4.0 / F)}
    from gnan_amd import _lib
    variants = [("tree", None, 0)]
    variants += [(f"index_fg32_bs1024_B{b}", b, 0) for b in (256, 512, 1024)]
    variants += [(f"index_fg32_bs512_B{b}", b, _lib.FPWL_INDEX_BS512) for b in (256, 512)]
    variants += [(f"index_fg16_B{b}", b, _lib.FPWL_INDEX_HALF_LINES) for b in (1024,)]
    for name, b, fl in variants:
        functional.INDEX_LOOKUP = b is not None
        functional.INDEX_FLAGS = fl
        if b is not None:
            functional.INDEX_BUCKETS = b
        row = {"variant": name, "nodes": N, "pieces_max": tb.max_pieces, "group_pieces": tb.max_group_pieces}
        for mode, (sumf, dt, bpl) in modes.items():
            with torch.no_grad():
                fn = lambda: functional._fpwl_launch(x, tb, sumf, want_total=not sumf, out_dtype=dt,
                                                     x_range=rng if b is not None else None)
                med, mn = timed(fn)
            row[mode + "_ms"] = round(med, 4)
            row[mode + "_frac_of_8TBps"] = round(N * F * bpl / (med / 1e3) / 8e12, 3)
        located = []
        with torch.no_grad():
            fn = lambda: functional._fpwl_launch(x, tb, True, located=located, x_range=rng if b is not None else None)
            row["sum_keep_pieces_ms"] = round(timed(fn)[0], 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
