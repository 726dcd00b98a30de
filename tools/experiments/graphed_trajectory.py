import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import microbench as mb
import graphed_step as gs
from gnan_amd import harness
DEV = "cuda"
d, n, F, C = gs.arxiv_shaped(1)
g = torch.Generator().manual_seed(1)
d.y = torch.randint(0, 2, (n,), generator=g).to(DEV)
r = torch.rand(n, generator=g)
d.train_mask, d.val_mask, d.test_mask = (r < 0.6).to(DEV), ((r >= 0.6) & (r < 0.8)).to(DEV), (r >= 0.8).to(DEV)
loss_fn = torch.nn.BCEWithLogitsLoss()
for tag, on in (("eager", False), ("graphed", True)):
    harness.GRAPHED_STEPS = on
    torch.manual_seed(0)
    m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device=DEV)
    mb.redraw(m)
    m = m.to(DEV).eval()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    ls = []
    for e in range(45):
        ls.append(round(harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False)[0], 5))
        if e % 3 == 0 and "noeval" not in sys.argv:
            if "eagereval" in sys.argv:
                harness.GRAPHED_STEPS = False
            harness.test_epoch(m, [d], loss_fn, DEV, classify=True, val_mask=True, is_graph_task=False)
            harness.GRAPHED_STEPS = on
    print(tag, ls)
    if on:
        for r in harness._STEPS[m].entries.values():
            print("replays", r.value["step"].graph.replays if r.value["step"] else None, r.value["optimizer"] is not None)
