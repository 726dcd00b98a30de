"""Debug aid: gradients / parameters per epoch of the harness's eager loop with gnan_amd.replay on vs off, per-layer optimizer."""
import copy, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnan_amd
from gnan_amd import harness, replay
from test_gpu_graphed import _node_task, _model, DEV
harness.GRAPHED_STEPS = False
data = _node_task(3000, 129, 1, False)
loss_fn = torch.nn.BCEWithLogitsLoss()
def make():
    m = _model(129, 1)
    fs, rho = list(m.fs.parameters()), list(m.rho.parameters())
    return m, torch.optim.Adam([{"params": fs}, {"params": rho, "lr": 3e-4}], lr=1e-3, capturable=True, fused=True)
a, oa = make(); b, ob = make()
for e in range(6):
    replay.REPLAY_FORWARD = False
    la = harness.train_epoch(a, [data], loss_fn, oa, DEV, classify=True, is_graph_task=False)[0]
    replay.REPLAY_FORWARD = True
    lb = harness.train_epoch(b, [data], loss_fn, ob, DEV, classify=True, is_graph_task=False)[0]
    ga = {k: p.grad for k, p in a.named_parameters()}; gb = {k: p.grad for k, p in b.named_parameters()}
    gs = max(float(v.abs().max()) for v in ga.values() if v is not None)
    worst = max(((float((ga[k] - gb[k]).abs().max()) / gs if (ga[k] is not None and gb[k] is not None) else (0.0 if ga[k] is gb[k] else 9.9)), k) for k in ga)
    ps = max(float(v.abs().max()) for v in a.state_dict().values())
    pw = max((float((va - vb).abs().max()) / ps, k) for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()))
    st = [float(ob.state[p]["step"]) for p in list(ob.state)[:1]], [float(oa.state[p]["step"]) for p in list(oa.state)[:1]]
    print(e, "loss", la, lb, "worst grad diff", worst, "worst param diff", pw, "steps", st, flush=True)
