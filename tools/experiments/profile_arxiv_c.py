"""arxiv-shaped forward+backward with C output channels (C in {1, 40}): kernel split from torch.profiler.
    python tools/experiments/profile_arxiv_c.py [C [nodes edges features]]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import microbench as mb
import torch
from torch.profiler import profile, ProfilerActivity
C = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N, E, F = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (169_343, 1_166_243, 129)
gen = torch.Generator(device="cuda").manual_seed(0)
src = torch.randint(0, N, (E,), generator=gen, device="cuda")
dst = (torch.rand(E, generator=gen, device="cuda") ** 3 * N).long().clamp_(0, N - 1)
g = mb.syn.hop1_csr(src, dst, N)
x = mb.syn.block_features(N, F, 0, N, 1, "cuda")
d = mb.Bag(x=x, edge_index=None, gnan_graph=g)
torch.manual_seed(0)
m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device="cuda")
mb.redraw(m)
m = m.to("cuda").eval()
def fb():
    m.zero_grad(set_to_none=True)
    m.forward(d).pow(2).sum().backward()
print("fwd+bwd ms (median, min):", mb.timeit(fb, reps=20, warm=3))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        fb()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = 0
for e in rows[:22]:
    if e.device_time_total > 0 and e.device_type.name != "CPU":
        print(f"{e.device_time_total / 5 / 1e3:8.3f} ms/step x{e.count / 5:5.1f}  {e.key[:110]}")
