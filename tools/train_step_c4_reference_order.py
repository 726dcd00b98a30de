import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gnan_amd
from gnan_amd import synthetic as syn
from gnan_amd.models import TensorGNAN
DEV="cuda"; N,E,SCALE=10_000_000,100_000_000,24; F=64
class Bag:
    def __init__(self, **kw): self.__dict__.update(kw)
torch.manual_seed(0)
src, dst = syn.rmat_edges(SCALE, N, E, seed=0, device=DEV)
g = syn.hop1_csr(src, dst, N); del src, dst
x = syn.block_features(N, F, 0, N, seed=1, device=DEV)
y = torch.randn(N, 1, device=DEV)
m = TensorGNAN(F, 1, 3, hidden_channels=64, device=DEV)
with torch.no_grad():
    for _, p in m.named_parameters():
        torch.nn.init.xavier_normal_(p, gain=1.0) if p.dim() == 2 else p.normal_(0.0, 0.5)
m = m.to(DEV).eval()
m.aggregation_order = "reference"
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
d = Bag(x=x, edge_index=None, gnan_graph=g)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.mse_loss(m.forward(d), y); loss.backward(); opt.step(); return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print("reference-order train step ms", (time.perf_counter()-t0)/5*1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]:
    print(f"{e.device_time_total/3/1e3:8.3f} ms/step x{e.count/3:4.1f} {e.key[:100]}")
