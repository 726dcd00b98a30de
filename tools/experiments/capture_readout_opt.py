import os, sys, faulthandler, copy
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import microbench as mb
from gnan_amd import HopGraph
from gnan_amd.graphed import GraphedStep
DEV = "cuda"
rng = np.random.default_rng(0)
F, n = 15, 12
readout = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def make():
    ei = np.stack([np.arange(1, n), rng.integers(0, np.arange(1, n))])
    g = HopGraph.from_edge_index(torch.as_tensor(np.concatenate([ei, ei[::-1]], 1)).to(DEV), n)
    x = torch.zeros(n, F); x[torch.arange(n), torch.from_numpy(rng.integers(0, F - 1, n))] = 1; x[:, -1] = 1
    return g, x.to(DEV)
graphs = [make() for _ in range(30)]
graphs = [(g, x) for g, x in graphs if g.n_codes == graphs[0][0].n_codes][:8]
print("graphs", len(graphs))
y = torch.ones(1, device=DEV)
def run(graphed):
    torch.manual_seed(0)
    m = mb.TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=readout, device=DEV)
    torch.manual_seed(1); mb.redraw(m); m = m.to(DEV).eval()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3)
    g0, x0 = graphs[0]
    xs, code, cnt = torch.empty_like(x0), torch.empty_like(g0.code), torch.empty_like(g0.cnt)
    static = mb.Bag(x=xs, edge_index=None, gnan_graph=HopGraph(n_rows=n, n_cols=n, n_codes=g0.n_codes, code=code, cnt=cnt))
    def loss_of(out):
        return torch.nn.functional.binary_cross_entropy_with_logits(out.flatten(), y), None
    step = None
    hist = []
    for rep in range(3):
        for i, (g, x) in enumerate(graphs):
            k = rep * len(graphs) + i
            if graphed and k >= 2:
                xs.copy_(x); code.copy_(g.code); cnt.copy_(g.cnt)
                if step is None:
                    step = GraphedStep(m, static, loss_of, opt, warmup=0)
                _, loss, _ = step.replay()
            else:
                opt.zero_grad(set_to_none=True)
                loss, _ = loss_of(m.forward(mb.Bag(x=x, edge_index=None, gnan_graph=g)))
                loss.backward(); opt.step()
                loss = loss.detach()
            torch.cuda.synchronize()
            hist.append((round(float(loss), 6), round(sum(float(p.double().abs().sum()) for p in m.parameters()), 5)))
            del loss
    return hist
a, b = run(False), run(True)
for k, (u, v) in enumerate(zip(a, b)):
    print(k, u, v, "" if abs(u[1] - v[1]) < 1e-3 * abs(u[1]) and abs(u[0] - v[0]) < 1e-4 * max(1, abs(u[0])) else "<-- DIFF")
