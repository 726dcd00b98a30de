"""EXPERIMENT: is a column-partitioned hot phase worth building?  

Only gathers that hit the 4-MiB L2 of their XCD are cheaper than HBM gathers (tools/gather_ceiling.hip).  The K hottest
operand rows (by in-degree) receive half of all pairs of the R-MAT graph but do not fit one L2 together; split over the 8
XCDs by rank they would (K = 65536: 2 MiB each).  This script measures, with the EXISTING kernel, the pairs of ONE
partition processed alone (every XCD then holds that partition's 2 MiB — the same per-L2 working set the real design
would have) and the cold remainder alone; 8 x hot + cold estimates the split kernel.
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnan_amd  # noqa
from gnan_amd import HopGraph, synthetic as syn
from gnan_amd.aggregate import spmm_launch

dev = torch.device("cuda")
N, E, W = 10_000_000, 100_000_000, 64
bf16 = "--bf16" in sys.argv
src, dst = syn.rmat_edges(24, N, E, seed=0, device=dev)
g = syn.hop1_csr(src, dst, N)
del src, dst
S = torch.rand((N, W), device=dev)
if bf16:
    S = S.bfloat16()
lut = torch.tensor([[0.7], [-0.3], [0.2]], device=dev)
total = torch.zeros(W, device=dev)
indeg = torch.bincount(g.col.long(), minlength=N)
order = torch.argsort(indeg, descending=True, stable=True)
rank = torch.empty(N, dtype=torch.int64, device=dev)
rank[order] = torch.arange(N, device=dev)
deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
row_of_pair = torch.repeat_interleave(torch.arange(N, device=dev), deg)


def sub(mask):
    cnt = torch.bincount(row_of_pair[mask], minlength=N)
    rp = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(cnt, 0)
    return HopGraph.from_csr(rp.to(torch.int32), g.col[mask], g.code[mask], n_cols=N, n_codes=3, cnt=g.cnt)


def bench(graph, label):
    for _ in range(3):
        y = spmm_launch(graph, S, lut, True, True, reduce_cr=1, s_total=total)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        y = spmm_launch(graph, S, lut, True, True, reduce_cr=1, s_total=total)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{label:46s} pairs {graph.nnz:>10d}  {ms:7.3f} ms  {graph.nnz / ms / 1e6:6.1f} G pairs/s", flush=True)
    return ms


t_all = bench(g, "whole graph")
for K in [int(k) for k in sys.argv[1:] if not k.startswith("--")] or [65536]:
    r = rank[g.col.long()]
    hot = r < K
    t_cold = bench(sub(~hot), f"K={K}: cold pairs alone")
    t_hot = [bench(sub(hot & (r % 8 == p)), f"K={K}: hot partition {p} alone") for p in (0, 3)]
    t_hot_all = bench(sub(hot), f"K={K}: all hot pairs, unpartitioned")
    est = t_cold + 8 * sum(t_hot) / len(t_hot)
    print(f"K={K}: estimate cold + 8 x hot partition = {est:.3f} ms   (whole graph {t_all:.3f} ms; cold + unpartitioned hot {t_cold + t_hot_all:.3f} ms)")
