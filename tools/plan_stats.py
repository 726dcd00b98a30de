"""Hub-row plan of the bench graph: how many rows exceed the threshold, how many slices, the widest row."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnan_amd  # noqa: F401
from gnan_amd import synthetic as syn

N, E, scale = 10_000_000, 100_000_000, 24
dev = torch.device("cuda", 0)
src, dst = syn.rmat_edges(scale, N, E, seed=0, device=dev)
g = syn.hop1_csr(src, dst, N)
deg = (g.rowptr[1:] - g.rowptr[:-1])
plan = g.long_row_plan()
ns = (plan.slice_ptr[1:] - plan.slice_ptr[:-1]) if plan.n_long else torch.zeros(1)
print(json.dumps({"n_long": plan.n_long, "n_slices": plan.n_slices, "max_deg": int(deg.max()),
                  "max_slices_per_row": int(ns.max()), "deg_gt_2048": int((deg > 2048).sum()),
                  "pairs_in_long_rows": int(deg[deg > 512].sum())}))
