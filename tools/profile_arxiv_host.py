"""Host-side profile (cProfile) of the arxiv-shaped forward and forward+backward (where does the Python time go?)."""
import cProfile, pstats, os, sys, io
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
def fwd():
    with torch.no_grad():
        return m.forward(d)
def fb():
    m.zero_grad(set_to_none=True)
    m.forward(d).pow(2).sum().backward()
for name, fn, reps in (("forward", fwd, 50), ("forward+backward", fb, 20)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(18)
    print("=====", name, "x", reps)
    print("\n".join(l[:150] for l in out.getvalue().splitlines()[:40]))
