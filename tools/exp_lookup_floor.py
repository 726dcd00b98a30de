"""Experiment: how much of the look-up kernel's time is the binary search (vs. pure streaming)?"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnan_amd
from gnan_amd import pwl, functional, synthetic as syn
from gnan_amd.functional import stack_mlps
from gnan_amd.models import TensorGNAN
dev = "cuda"
N, F = 10_000_000, 64
m = TensorGNAN(F, 1, 3, hidden_channels=64, device=dev)
with torch.no_grad():
    for p in m.parameters():
        (torch.nn.init.xavier_normal_(p, gain=1.0) if p.dim() == 2 else p.normal_(0, 0.5))
m = m.to(dev).eval()
x = syn.block_features(N, F, 0, N, 1, dev)
with torch.no_grad():
    tb = pwl.build_tables(stack_mlps(m.fs))
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("full search      ", t(lambda: functional._fpwl_launch(x, tb, False)))
for mp in (65, 17, 5, 1):
    tb2 = tb._replace(max_pieces=mp)     # fewer search steps (results wrong; timing only)
    print(f"max_pieces={mp:3d}   ", t(lambda: functional._fpwl_launch(x, tb2, False)))
y = torch.empty_like(x)
print("copy x->y (torch)", t(lambda: y.copy_(x)))
