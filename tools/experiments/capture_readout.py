import os, sys, faulthandler
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import microbench as mb
from gnan_amd import HopGraph
from gnan_amd.graphed import GraphedCallable
DEV = "cuda"
rng = np.random.default_rng(0)
F, n = 15, int(sys.argv[2]) if len(sys.argv) > 2 else 12
readout = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def make():
    ei = np.stack([np.arange(1, n), rng.integers(0, np.arange(1, n))])
    g = HopGraph.from_edge_index(torch.as_tensor(np.concatenate([ei, ei[::-1]], 1)).to(DEV), n)
    x = torch.zeros(n, F); x[torch.arange(n), torch.from_numpy(rng.integers(0, F - 1, n))] = 1; x[:, -1] = 1
    return g, x.to(DEV)
graphs = [make() for _ in range(6)]
graphs = [(g, x) for g, x in graphs if g.n_codes == graphs[0][0].n_codes]
print("same-shape graphs:", len(graphs), "D", graphs[0][0].n_codes)
torch.manual_seed(0)
m = mb.TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=readout, device=DEV)
torch.manual_seed(1); mb.redraw(m); m = m.to(DEV).eval()
y = torch.ones(1, device=DEV)
g0, x0 = graphs[0]
xs, code, cnt = torch.empty_like(x0), torch.empty_like(g0.code), torch.empty_like(g0.cnt)
static = mb.Bag(x=xs, edge_index=None, gnan_graph=HopGraph(n_rows=n, n_cols=n, n_codes=g0.n_codes, code=code, cnt=cnt))
def step(d):
    out = m.forward(d)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out.flatten(), y)
    loss.backward()
    return loss.detach()
def sig():
    torch.cuda.synchronize()
    return [round(float(p.grad.double().abs().sum()), 6) for p in list(m.parameters())[::29]]
ref = []
for g, x in graphs:
    m.zero_grad(set_to_none=True)
    l = step(mb.Bag(x=x, edge_index=None, gnan_graph=g))
    ref.append((float(l), sig()))
m.zero_grad(set_to_none=True)
xs.copy_(x0); code.copy_(g0.code); cnt.copy_(g0.cnt)
gc = GraphedCallable(lambda: step(static), warmup=0, before_capture=lambda: m.zero_grad(set_to_none=True))
for rep in range(2):
    for i, (g, x) in enumerate(graphs):
        xs.copy_(x); code.copy_(g.code); cnt.copy_(g.cnt)
        gc.replay()
        s = sig()
        print(rep, i, "loss", round(float(gc.out), 6), round(ref[i][0], 6), "OK" if s == ref[i][1] else f"DIFF {s} vs {ref[i][1]}")
