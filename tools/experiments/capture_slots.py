import os, sys, faulthandler
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import microbench as mb
from gnan_amd import harness, graphed
DEV = "cuda"
rng = np.random.default_rng(0)
F = 15
graphs = []
for i in range(12):
    n = [12, 30, 12][i % 3]
    ei = np.stack([np.arange(1, n), rng.integers(0, np.arange(1, n))])
    nd, norm = mb.dense_inputs(np.concatenate([ei, ei[::-1]], 1), n)
    x = torch.zeros(n, F); x[torch.arange(n), torch.from_numpy(rng.integers(0, F - 1, n))] = 1; x[:, -1] = 1
    y = torch.tensor([[1.0 if rng.random() < 0.5 else -1.0]])
    class D(mb.Bag):
        def to(self, device): return self
    graphs.append(D(x=x.to(DEV), y=y.to(DEV), edge_index=None, node_distances=nd, normalization_matrix=norm))
torch.manual_seed(0)
m = mb.TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=0, device=DEV)
mb.redraw(m); m = m.to(DEV).eval()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
loss_fn = torch.nn.BCEWithLogitsLoss()
orig = graphed.SlottedGraphStep.__init__
def traced(self, *a, **k):
    print("capturing shape", a[3].n_rows, a[3].n_codes, flush=True)
    orig(self, *a, **k)
    print("captured; memsets replaced:", self.step.graph.memsets_replaced, flush=True)
graphed.SlottedGraphStep.__init__ = traced
harness.GRAPHED_STEPS = True
for e in range(3):
    print("epoch", e, harness.train_epoch(m, graphs, loss_fn, opt, DEV, classify=True, is_graph_task=True), flush=True)
