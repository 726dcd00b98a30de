"""Experiment: does processing rows in degree-sorted order help the narrow-row aggregation?"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnan_amd
from gnan_amd import synthetic as syn
from gnan_amd.functional import column_sums
from gnan_amd.aggregate import spmm_launch

dev = "cuda"
N, E = 10_000_000, 100_000_000
src, dst = syn.rmat_edges(24, N, E, 0, dev)
g = syn.hop1_csr(src, dst, N)
del src, dst
g.long_row_plan()
deg = (g.rowptr[1:] - g.rowptr[:-1])
order = torch.argsort(deg, stable=True).to(torch.int32)
lut = torch.tensor([[1.0], [0.5], [0.01]], device=dev)

def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for W in (1, 4, 8, 16, 64):
    S = torch.rand(N, W, device=dev)
    tot = column_sums(S)
    rc = 1 if W > 1 else 0
    a = t(lambda: spmm_launch(g, S, lut, True, True, s_total=tot, reduce_cr=rc))
    b = t(lambda: spmm_launch(g, S, lut, True, True, row_ids=order, s_total=tot, reduce_cr=rc))
    y0 = spmm_launch(g, S, lut, True, True, s_total=tot, reduce_cr=rc)
    y1 = spmm_launch(g, S, lut, True, True, row_ids=order, s_total=tot, reduce_cr=rc)
    ok = torch.equal(y0[order.long()], y1)
    print(f"W={W}: natural {a:.3f} ms, degree-sorted {b:.3f} ms (excl. un-permute), equal={ok}", flush=True)
