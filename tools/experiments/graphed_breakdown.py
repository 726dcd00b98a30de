import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import microbench as mb
import graphed_step as gs
from gnan_amd import harness
DEV = "cuda"
d, n, F, C = gs.arxiv_shaped(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
g = torch.Generator().manual_seed(1)
d.y = torch.randint(0, max(C, 2), (n,), generator=g).to(DEV)
r = torch.rand(n, generator=g)
d.train_mask, d.val_mask, d.test_mask = (r < 0.6).to(DEV), ((r >= 0.6) & (r < 0.8)).to(DEV), (r >= 0.8).to(DEV)
loss_fn = torch.nn.BCEWithLogitsLoss() if C == 1 else torch.nn.CrossEntropyLoss()
harness.GRAPHED_STEPS = True
torch.manual_seed(0)
m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device=DEV)
mb.redraw(m)
m = m.to(DEV).eval()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
for _ in range(5):
    harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False)
rec = [r.value for r in harness._STEPS[m].entries.values()][0]
step = rec["step"]
def t(fn, reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print(json.dumps({"stale_ms": t(step.stale), "fits_ms": t(step.graph.fits), "graph_replay_ms": t(step.graph.graph.replay),
                  "step_replay_ms": t(step.replay),
                  "epoch_ms": t(lambda: harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False))}))
