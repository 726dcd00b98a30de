import os, sys, faulthandler
faulthandler.enable()
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import test_gpu_graphed as T
from gnan_amd import harness
from gnan_amd.models import TensorGNAN
DEV = "cuda"
F = 15
readout = int(sys.argv[1]); n_graphs = int(sys.argv[2]); per_call = int(sys.argv[3])
graphs = T._graph_task(n_graphs, F, sizes=[12, 30, 12, 23, 30, 12, 41])
if "noiso" in sys.argv:
    graphs = [g for i, g in enumerate(graphs) if i % 5 != 0]
loss_fn = torch.nn.BCEWithLogitsLoss()
def run(on, fused=False):
    harness.GRAPHED_STEPS = on
    torch.manual_seed(0)
    m = TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=readout, device=DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    m = m.to(DEV).eval()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3, capturable=fused, fused=fused) if fused else torch.optim.Adam(m.parameters(), lr=2e-3)
    hist = []
    for rep in range(4):
        for i in range(0, len(graphs), per_call):
            l = harness.train_epoch(m, graphs[i:i + per_call], loss_fn, opt, DEV, classify=True, is_graph_task=True)[0]
            hist.append((round(l, 6), round(sum(float(p.detach().double().abs().sum()) for p in m.parameters()), 5)))
    return hist
a, b = run(False), (run(False, fused=True) if "fusedeager" in sys.argv else run(True))
for k, (u, v) in enumerate(zip(a, b)):
    bad = not (abs(u[1] - v[1]) < 1e-5 * abs(u[1]) and abs(u[0] - v[0]) < 1e-4 * max(1, abs(u[0])))
    if bad:
        print("first diff at call", k, "graph", (k * per_call) % len(graphs), "n", graphs[(k * per_call) % len(graphs)].x.shape[0], u, v)
        break
else:
    print("no diff over", len(a), "calls")
