import os, sys, faulthandler
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import microbench as mb
from gnan_amd import harness
DEV = "cuda"
rng = np.random.default_rng(0)
F = 15
readout = int(sys.argv[1]) if len(sys.argv) > 1 else 2
class D(mb.Bag):
    def to(self, device): return self
def make(n):
    ei = np.stack([np.arange(1, n), rng.integers(0, np.arange(1, n))])
    nd, norm = mb.dense_inputs(np.concatenate([ei, ei[::-1]], 1), n)
    x = torch.zeros(n, F); x[torch.arange(n), torch.from_numpy(rng.integers(0, F - 1, n))] = 1; x[:, -1] = 1
    y = torch.tensor([[1.0 if rng.random() < 0.5 else -1.0]])
    return D(x=x.to(DEV), y=y.to(DEV), edge_index=None, node_distances=nd, normalization_matrix=norm)
graphs = [make([12, 20][i % 2]) for i in range(10)]
loss_fn = torch.nn.BCEWithLogitsLoss()
def run(on):
    harness.GRAPHED_STEPS = on
    torch.manual_seed(0)
    m = mb.TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=readout, device=DEV)
    torch.manual_seed(1); mb.redraw(m); m = m.to(DEV).eval()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3)
    hist = []
    for rep in range(4):
        for g in graphs:
            l = harness.train_epoch(m, [g], loss_fn, opt, DEV, classify=True, is_graph_task=True)[0]
            hist.append((round(l, 6), round(sum(float(p.detach().double().abs().sum()) for p in m.parameters()), 5)))
    if on:
        st = harness._GRAPH_STEPS[m]
        print("buckets", {k[:3]: (r["calls"], r["step"] is not None and r["step"].step.graph.replays) for k, r in st.buckets.items()})
    return hist
a, b = run(False), run(True)
for k, (u, v) in enumerate(zip(a, b)):
    bad = not (abs(u[1] - v[1]) < 1e-6 * abs(u[1]) and abs(u[0] - v[0]) < 1e-4 * max(1, abs(u[0])))
    if bad or k < 3:
        print(k, u, v, "<-- DIFF" if bad else "")
    if bad:
        break
