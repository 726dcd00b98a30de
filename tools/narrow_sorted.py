"""EXPERIMENT: narrow operand rows (W = 1..4) walked in degree order through the degree-sorted copy of the CSR
(aggregate.NARROW_SORTED_WALK), with and without the hot rows appended (aggregate.HOT_COLUMN_ROWS), against natural order.     python tools/experiments/narrow_sorted.py [W ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnan_amd  # noqa
from gnan_amd import functional, synthetic as syn
from gnan_amd.functional import column_sums
from gnan_amd.aggregate import spmm_launch

dev = torch.device("cuda")
N, E = 10_000_000, 100_000_000
src, dst = syn.rmat_edges(24, N, E, seed=0, device=dev)
g = syn.hop1_csr(src, dst, N)
del src, dst
lut = torch.tensor([[0.7], [-0.3], [0.2]], device=dev)


def bench(op, total, label):
    for _ in range(3):
        y = spmm_launch(g, op, lut, True, True, s_total=total)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        y = spmm_launch(g, op, lut, True, True, s_total=total)
    b.record()
    torch.cuda.synchronize()
    print(f"{label:50s} {a.elapsed_time(b) / 10:.3f} ms  checksum {float(y.double().sum()):.6f}", flush=True)
    return y


for W in [int(w) for w in sys.argv[1:]] or [1, 2, 4]:
    S = torch.rand((N, W), device=dev)
    total = column_sums(S)
    ys = []
    for label, walk, hot in (("natural order", False, False), ("degree-sorted copy", True, False),
                             ("degree-sorted copy + hot rows appended", True, True)):
        aggregate.NARROW_SORTED_WALK, aggregate.HOT_COLUMN_ROWS = walk, hot
        ys.append(bench(S, total, f"W = {W}: {label}"))
    print("    identical:", all(bool(torch.equal(ys[0], y)) for y in ys[1:]))
