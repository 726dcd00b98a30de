"""EXPERIMENT (narrow operands): a compact copy of the hottest operand rows appended to the operand.

Narrow operand rows (sum-first, W = C <= 4: 4..16 bytes) sit on the L2-miss request rate (110M gathers = 1.9 ms on the
10M-node graph).  The K highest in-degree nodes receive a large share of the pairs, but their 4-byte values are spread
over K different 128-B lines that do not survive in L2.  Here their values are copied to K extra operand rows
[n, n + K) — 32 hot nodes per line at W = 1 — and the column ids of the pairs that list them point there: no kernel
change, one extra int32 column array per graph and a K-row gather per forward.

    python tools/experiments/hot_columns.py [W] [K ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnan_amd  # noqa
from gnan_amd import HopGraph, synthetic as syn
from gnan_amd import functional
from gnan_amd.functional import column_sums
from gnan_amd.aggregate import spmm_launch

aggregate.DEGREE_SCHEDULE_MIN_WIDTH = int(os.environ.get("MIN_WIDTH", "8"))

dev = torch.device("cuda")
N, E = 10_000_000, 100_000_000
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Ks = [int(k) for k in sys.argv[2:]] or [16384, 65536, 262144, 1048576]
src, dst = syn.rmat_edges(24, N, E, seed=0, device=dev)
g = syn.hop1_csr(src, dst, N)
del src, dst
S = torch.rand((N, W), device=dev)
lut = torch.tensor([[0.7], [-0.3], [0.2]], device=dev)
total = column_sums(S)
indeg = torch.bincount(g.col.long(), minlength=N)
order = torch.argsort(indeg, descending=True, stable=True)        # order[rank] = node
csum = torch.cumsum(indeg[order], 0)


def bench(graph, op, label):
    for _ in range(3):
        y = spmm_launch(graph, op, lut, True, True, s_total=total)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        y = spmm_launch(graph, op, lut, True, True, s_total=total)
    b.record()
    torch.cuda.synchronize()
    print(f"{label:60s} {a.elapsed_time(b) / 10:.3f} ms  checksum {float(y.double().sum()):.6f}", flush=True)
    return y


y0 = bench(g, S, f"W = {W}: natural numbering")
for K in Ks:
    hot = order[:K]
    rank = torch.full((N,), -1, dtype=torch.int64, device=dev)
    rank[hot] = torch.arange(K, device=dev)
    r = rank[g.col.long()]
    col2 = torch.where(r >= 0, r + N, g.col.long()).to(torch.int32)
    g2 = HopGraph.from_csr(g.rowptr, col2, g.code, n_cols=N + K, n_codes=3, cnt=g.cnt)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    S2 = torch.empty((N + K, W), device=dev)
    S2[:N] = S
    torch.cuda.synchronize()
    a.record()
    S2[N:] = S[hot]
    b.record()
    torch.cuda.synchronize()
    y = bench(g2, S2, f"hottest {K} rows appended ({float(csum[K - 1]) / float(csum[-1]):.1%} of the pairs, copy {a.elapsed_time(b):.3f} ms)")
    print("    max |diff|", float((y0 - y).abs().max()))
    del g2, S2, col2, r, rank
