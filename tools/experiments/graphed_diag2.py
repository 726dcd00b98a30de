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

def run(graph_eval):
    harness.GRAPHED_STEPS = True
    torch.manual_seed(0)
    m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device=DEV)
    mb.redraw(m)
    m = m.to(DEV).eval()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for e in range(5):
        harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False)
    for i in range(4):
        harness.GRAPHED_STEPS = graph_eval
        harness.test_epoch(m, [d], loss_fn, DEV, classify=True, val_mask=True, is_graph_task=False)
        harness.GRAPHED_STEPS = True
    harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False)
    torch.cuda.synchronize()
    snap = {}
    for name, p in m.named_parameters():
        snap["p/" + name] = p.detach().clone()
        snap["g/" + name] = p.grad.detach().clone()
        for k in ("exp_avg", "exp_avg_sq", "step"):
            snap[k + "/" + name] = opt.state[p][k].detach().clone()
    return snap

a = run(False)
b = run(True)
bad = {}
for k in a:
    dlt = float((a[k].double() - b[k].double()).abs().max())
    if dlt > 0:
        kind, name = k.split("/", 1)
        grp = name.split(".")[0] + ("." + name.split(".")[2] if name.startswith("fs.") else "")
        bad.setdefault((kind, grp), []).append((name, dlt, float(a[k].double().abs().max())))
for key, v in sorted(bad.items()):
    v.sort(key=lambda t: -t[1])
    print(key, len(v), v[:2])
print("total tensors compared", len(a))
