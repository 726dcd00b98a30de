import os, sys, subprocess
if len(sys.argv) < 2:
    for mode in ("fwd", "fwdbwd", "step"):
        for keep in ("keep", "nokeep"):
            r = subprocess.run([sys.executable, __file__, mode, keep], capture_output=True, text=True)
            print(mode, keep, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1][:200], flush=True)
    sys.exit(0)
mode, keep = sys.argv[1], sys.argv[2] == "keep"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import microbench as mb
from gnan_amd import functional
DEV = "cuda"
rng = np.random.default_rng(0)
n, F = 12, 15
ei = np.stack([np.arange(1, n), rng.integers(0, np.arange(1, n))])
nd, norm = mb.dense_inputs(np.concatenate([ei, ei[::-1]], 1), n)
x = torch.zeros(n, F); x[torch.arange(n), torch.from_numpy(rng.integers(0, F - 1, n))] = 1; x[:, -1] = 1
d = mb.Bag(x=x.to(DEV), edge_index=None, node_distances=nd, normalization_matrix=norm)
m = mb.TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=0, device=DEV)
mb.redraw(m); m = m.to(DEV).eval()
opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, fused=True)
y = torch.ones(1, device=DEV)
def fn():
    if mode == "fwd":
        with torch.no_grad():
            return m.forward(d)
    out = m.forward(d)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out.flatten(), y)
    loss.backward()
    if mode == "step":
        opt.step()
    return loss.detach()
for _ in range(2):
    opt.zero_grad(set_to_none=True); fn()
torch.cuda.synchronize()
opt.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fn()
if keep:
    g.instantiate()
g.replay(); torch.cuda.synchronize()
print("ok", float(out.flatten()[0]))
