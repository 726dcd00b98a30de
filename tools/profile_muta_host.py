"""Host-side profile (cProfile) of the Mutagenicity-shaped per-graph forward and forward+backward."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench as mb
import numpy as np, torch

rng = np.random.default_rng(0)
graphs = []
for _ in range(50):
    n = int(np.clip(round(rng.lognormal(3.3, 0.45)), 4, 417))
    par = np.array([rng.integers(0, i) for i in range(1, n)])
    ei = np.stack([np.arange(1, n), par])
    ei = np.concatenate([ei, ei[::-1]], 1)
    nd, norm = mb.dense_inputs(ei, n)
    x = torch.zeros(n, 15); x[torch.arange(n), torch.from_numpy(rng.integers(0, 14, n))] = 1; x[:, -1] = 1
    graphs.append(mb.Bag(x=x.to("cuda"), edge_index=torch.from_numpy(ei).to("cuda"), node_distances=nd.to("cuda"),
                         normalization_matrix=norm.to("cuda")))
m = mb.TensorGNAN(15, 1, 3, hidden_channels=64, is_graph_task=True, readout_n_layers=0, device="cuda")
mb.redraw(m); m = m.to("cuda").eval()
opt = torch.optim.SGD(m.parameters(), lr=0.0)
def fwd():
    with torch.no_grad():
        for g in graphs: m.forward(g)
def fb():
    for g in graphs:
        opt.zero_grad(set_to_none=True); m.forward(g).pow(2).sum().backward()
for name, fn, reps in (("forward", fwd, 4), ("forward+backward", fb, 2)):
    fn(); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); pr.disable()
    out = io.StringIO(); pstats.Stats(pr, stream=out).strip_dirs().sort_stats("tottime").print_stats(22)
    print("=====", name, "x", reps * len(graphs), "graphs")
    print("\n".join(l[:160] for l in out.getvalue().splitlines()[:36]))
