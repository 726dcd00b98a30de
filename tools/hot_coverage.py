"""Share of the listed pairs of the C4 graph whose neighbour is among the K most listed nodes (what a compact / LDS-resident
copy of the hottest operand rows can serve)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gnan_amd  # noqa
from gnan_amd import synthetic as syn
N, E = 10_000_000, 100_000_000
src, dst = syn.rmat_edges(24, N, E, seed=0, device="cuda")
listed = torch.bincount(dst, minlength=N) + 1          # + the self pair
top = torch.sort(listed, descending=True).values.double()
cum = torch.cumsum(top, 0) / top.sum()
out = {str(k): round(float(cum[k - 1]), 4) for k in (1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304)}
deg = torch.bincount(src, minlength=N) + 1
print(json.dumps({"coverage_by_top_k_neighbours": out, "max_listed": int(top[0]), "max_degree": int(deg.max()),
                  "rows_deg_le_4": float((deg <= 4).double().mean()), "mean_deg": float(deg.double().mean())}))
