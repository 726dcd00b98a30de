import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import microbench as mb
import graphed_step as gs
from gnan_amd import harness, functional
DEV = "cuda"
d, n, F, C = gs.arxiv_shaped(1)
g = torch.Generator().manual_seed(1)
d.y = torch.randint(0, 2, (n,), generator=g).to(DEV)
r = torch.rand(n, generator=g)
d.train_mask, d.val_mask, d.test_mask = (r < 0.6).to(DEV), ((r >= 0.6) & (r < 0.8)).to(DEV), (r >= 0.8).to(DEV)
loss_fn = torch.nn.BCEWithLogitsLoss()
CALLS = []
orig = functional.spmm_launch
def wrap(g_, S, lut, use_cnt, with_rest, row_ids=None, **kw):
    out = orig(g_, S, lut, use_cnt, with_rest, row_ids, **kw)
    if torch.cuda.is_current_stream_capturing():
        CALLS.append(dict(S=S, lut=lut, out=out, kw={k: (v if not torch.is_tensor(v) else "tensor") for k, v in kw.items()},
                          s_total=kw.get("s_total")))
    return out
functional.spmm_launch = wrap
origm = functional._fpwl_moments
G = {}
def wrapm(x, t, grad, sf, xam=None, raw=False):
    if torch.cuda.is_current_stream_capturing():
        G["grad"] = grad
    return origm(x, t, grad, sf, xam, raw)
functional._fpwl_moments = wrapm

def show(tag):
    torch.cuda.synchronize()
    print(tag, "dS-final absmax", float(G["grad"].abs().max()) if G else None)
    for i, c in enumerate(CALLS):
        st = c["s_total"]
        print("   call", i, c["kw"], "S", tuple(c["S"].shape), float(c["S"].double().abs().sum()), "lut", c["lut"].flatten().tolist()[:4],
              "out", float(c["out"].double().abs().sum()), "s_total", None if st is None else st.flatten().tolist()[:3])

harness.GRAPHED_STEPS = True
torch.manual_seed(0)
m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device=DEV)
mb.redraw(m)
m = m.to(DEV).eval()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
for e in range(5):
    harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False)
    if e >= 2:
        show(f"train {e}")
