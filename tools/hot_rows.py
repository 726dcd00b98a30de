"""EXPERIMENT: do hot operand rows stay in L2 if the cold ones are gathered with non-temporal loads?

Operand rows are renumbered by in-degree (hot first); an experimental build of the library (GNAN_HIP_LIB=…/libgnan_hip_nt.so,
csrc patch kept in tools/experiments/spmm_nt.patch) gathers rows with id >= GNAN_SPMM_NT_FROM with `nt` loads.
"""
import os
import sys
import time

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
lut = torch.tensor([[0.7], [-0.3], [0.2]], device=dev)
indeg = torch.bincount(g.col.long(), minlength=N)
order = torch.argsort(indeg, descending=True, stable=True)        # order[rank] = node
newid = torch.empty(N, dtype=torch.int64, device=dev)
newid[order] = torch.arange(N, device=dev)
g2 = HopGraph.from_csr(g.rowptr, newid[g.col.long()].to(torch.int32), g.code, n_cols=N, n_codes=3, cnt=g.cnt)
S2 = S[order].contiguous()
if bf16:
    S, S2 = S.bfloat16(), S2.bfloat16()


def bench(graph, op, label):
    for _ in range(3):
        y = spmm_launch(graph, op, lut, True, True, reduce_cr=1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        y = spmm_launch(graph, op, lut, True, True, reduce_cr=1)
    b.record()
    torch.cuda.synchronize()
    print(f"{label:40s} {a.elapsed_time(b) / 10:.3f} ms  checksum {float(y.double().sum()):.6f}", flush=True)
    return y


os.environ.pop("GNAN_SPMM_NT_FROM", None)
y0 = bench(g, S, "natural numbering")
y1 = bench(g2, S2, "renumbered by in-degree, no nt")
print("max |diff|", float((y0 - y1).abs().max()))
for k in sys.argv[1:]:
    if k.startswith("--"):
        continue
    os.environ["GNAN_SPMM_NT_FROM"] = k
    bench(g2, S2, f"renumbered, nt from row {k}")
os.environ["GNAN_SPMM_NT_FROM"] = "0"
bench(g, S, "natural numbering, all nt")
