"""arxiv-shaped forward+backward only (BASELINE.json config 2) — run under rocprofv3 for a kernel-level breakdown."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench as mb
import torch
N, E, F = 169_343, 1_166_243, 129
gen = torch.Generator(device="cuda").manual_seed(0)
src = torch.randint(0, N, (E,), generator=gen, device="cuda")
dst = (torch.rand(E, generator=gen, device="cuda") ** 3 * N).long().clamp_(0, N - 1)
g = mb.syn.hop1_csr(src, dst, N)
x = mb.syn.block_features(N, F, 0, N, 1, "cuda")
d = mb.Bag(x=x, edge_index=None, gnan_graph=g)
m = mb.TensorGNAN(F, 1, 3, hidden_channels=64, device="cuda")
mb.redraw(m)
m = m.to("cuda").eval()
def fb():
    m.zero_grad(set_to_none=True)
    m.forward(d).pow(2).sum().backward()
print("fwd+bwd ms (median, min):", mb.timeit(fb, reps=20, warm=3))
