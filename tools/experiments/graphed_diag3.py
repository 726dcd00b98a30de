import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import microbench as mb
import graphed_step as gs
from gnan_amd import harness, functional
DEV = "cuda"
d, n, F, C = gs.arxiv_shaped(1)
g = torch.Generator().manual_seed(1)
d.y = torch.randint(0, 2, (n,), generator=g).to(DEV)
r = torch.rand(n, generator=g)
d.train_mask, d.val_mask, d.test_mask = (r < 0.6).to(DEV), ((r >= 0.6) & (r < 0.8)).to(DEV), (r >= 0.8).to(DEV)
loss_fn = torch.nn.BCEWithLogitsLoss()
STASH = {}
orig_m, orig_p = functional._fpwl_moments, functional._fpwl_param_grads_launch
def wrap_m(x, t, grad, sf, xam=None, raw=False):
    out = orig_m(x, t, grad, sf, xam, raw)
    STASH.update(grad=grad, xam=xam, Mi=out[0] if raw else out, scales=out[1] if raw else None, off=t.off, anchor=t.anchor)
    return out
def wrap_p(params, t, moments, L, H, C, F):
    outs = orig_p(params, t, moments, L, H, C, F)
    STASH.update(outs=outs)
    return outs
functional._fpwl_moments, functional._fpwl_param_grads_launch = wrap_m, wrap_p

def show(tag):
    torch.cuda.synchronize()
    s = STASH
    print(tag, "grad absmax", float(s["grad"].abs().max()), "grad sum", float(s["grad"].double().sum()), "xam", float(s["xam"]),
          "scales", s["scales"].tolist(), "Mi abs sum", float(s["Mi"].double().abs().sum()),
          "off[F]", int(s["off"][-1]), "anchor absmax(valid)", float(s["anchor"][: int(s["off"][-1])].abs().max()),
          "outs", [float(o.double().abs().sum()) for o in s["outs"] if o is not None])

harness.GRAPHED_STEPS = True
torch.manual_seed(0)
m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device=DEV)
mb.redraw(m)
m = m.to(DEV).eval()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
for e in range(5):
    harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False)
    show(f"train {e}")
for i in range(3):
    harness.test_epoch(m, [d], loss_fn, DEV, classify=True, val_mask=True, is_graph_task=False)
for e in range(3):
    harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False)
    show(f"train after eval {e}")
