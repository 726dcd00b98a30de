"""Host-side profile (cProfile) of one Cora-shaped training step."""
import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench as mb
import numpy as np, torch
rng = np.random.default_rng(0)
n, F, C = 2708, 1434, 7
ei, (nd, norm) = mb.dense_graph(n, 3.9, rng)
x = torch.rand(n, F); x = x / x.sum(1, keepdim=True); x[:, -1] = 1
d = mb.Bag(x=x.to("cuda"), edge_index=None, node_distances=nd.to("cuda"), normalization_matrix=norm.to("cuda"))
m = mb.GNAN(F, C, num_layers=3, hidden_channels=64, device="cuda"); mb.redraw(m); m = m.to("cuda").eval()
def fb():
    m.zero_grad(set_to_none=True)
    m.forward(d).pow(2).sum().backward()
for _ in range(3): fb()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): fb()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
