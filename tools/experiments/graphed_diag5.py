import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import microbench as mb
import graphed_step as gs
from gnan_amd import harness, functional
from gnan_amd.graphed import GraphedCallable
DEV = "cuda"
d, n, F, C = gs.arxiv_shaped(1)
g = torch.Generator().manual_seed(1)
y = torch.randint(0, 2, (n,), generator=g).float().to(DEV)
idx = (torch.rand(n, generator=g) < 0.6).nonzero().flatten().to(DEV)
ym = y[idx]
loss_fn = torch.nn.BCEWithLogitsLoss()
torch.manual_seed(0)
m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device=DEV)
mb.redraw(m)
m = m.to(DEV).eval()
STASH = "stash" in sys.argv
KEEP = []
if STASH:
    origm = functional._fpwl_moments
    def wrapm(x, t, grad, sf, xam=None, raw=False):
        KEEP.append(grad)
        return origm(x, t, grad, sf, xam, raw)
    functional._fpwl_moments = wrapm

def step():
    out = m.forward(d)
    loss = loss_fn(out.index_select(0, idx).flatten(), ym)
    loss.backward()
    return loss.detach()

def sig():
    torch.cuda.synchronize()
    st = m._stores["fs"]
    return [round(float(v.double().abs().sum()), 6) for v in st.grad.values()] + [round(float(p.grad.double().abs().sum()), 8) for p in m.rho.parameters()]

for _ in range(2):
    m.zero_grad(set_to_none=True)
    step()
print("eager      ", sig())
gc = GraphedCallable(step, warmup=0, before_capture=lambda: m.zero_grad(set_to_none=True))
for i in range(3):
    gc.replay()
    print("replay", i, "  ", sig(), float(gc.out))
junk = torch.empty(1 << 28, device=DEV).normal_()
del junk
with torch.no_grad():
    m.forward(d)
for i in range(3, 6):
    gc.replay()
    print("replay", i, "  ", sig(), float(gc.out))
