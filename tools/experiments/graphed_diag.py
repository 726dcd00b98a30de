import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import microbench as mb
import graphed_step as gs
from gnan_amd import harness
DEV = "cuda"
d, n, F, C = gs.arxiv_shaped(1)
g = torch.Generator().manual_seed(1)
d.y = torch.randint(0, 2, (n,), generator=g).to(DEV)
r = torch.rand(n, generator=g)
d.train_mask, d.val_mask, d.test_mask = (r < 0.6).to(DEV), ((r >= 0.6) & (r < 0.8)).to(DEV), (r >= 0.8).to(DEV)
loss_fn = torch.nn.BCEWithLogitsLoss()
harness.GRAPHED_STEPS = True
torch.manual_seed(0)
m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device=DEV)
mb.redraw(m)
m = m.to(DEV).eval()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)

def sums():
    torch.cuda.synchronize()
    out = {"params": sum(float(p.double().sum()) for p in m.parameters())}
    for k in ("exp_avg", "exp_avg_sq", "step"):
        out[k] = sum(float(opt.state[p][k].double().sum()) for p in m.parameters() if p in opt.state)
    out["grads"] = sum(float(p.grad.double().sum()) for p in m.parameters() if p.grad is not None)
    store = m._stores["fs"]
    out["flatgrad"] = sum(float(v.double().sum()) for v in store.grad.values())
    return out

def diff(a, b):
    return {k: (a[k], b[k]) for k in a if a[k] != b[k]}

for e in range(5):
    harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False)
s0 = sums()
for i in range(4):
    harness.test_epoch(m, [d], loss_fn, DEV, classify=True, val_mask=True, is_graph_task=False)
    s1 = sums()
    print("eval call", i, "changed:", diff(s0, s1))
    s0 = s1
rec = [r.value for r in harness._STEPS[m].entries.values() if r.value["optimizer"] is not None][0]
step = rec["step"]
# what does one train replay do to the eval graph's static outputs and vice versa
erec = [r.value for r in harness._STEPS[m].entries.values() if r.value["optimizer"] is None][0]
ev = erec["step"]
print("eval outputs before train replay", float(ev.outputs.double().sum()), float(ev.loss))
o = step.replay(); torch.cuda.synchronize()
print("train loss", float(o[1]))
print("eval outputs after train replay ", float(ev.outputs.double().sum()), float(ev.loss))
a = sums()
ev.replay(); torch.cuda.synchronize()
b = sums()
print("eval replay changed:", diff(a, b))
print("ptr ranges: train outputs", step.outputs.data_ptr(), "eval outputs", ev.outputs.data_ptr())
