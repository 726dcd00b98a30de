"""f-3 on the GPU: harness.train_epoch / test_epoch driving the gnan_amd modules against numbers captured from the
REFERENCE's trainer.train_epoch / test_epoch driving the REFERENCE's GNAN classes (tests/golden/make_golden.py, cases
310-317: inputs, initial state_dict, per-epoch returns, updated state_dict) — SGD steps and Adam runs long enough for the
harness to capture the step into a hipGraph and replay it; and f-4: the interpretability exports from the kernels' tables."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, golden_names
from oracle import gnan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    batches = []
    for b in range(meta["n_batches"]):
        batches.append(Bag(**{k.split("/", 1)[1]: torch.from_numpy(z[k]).to(DEV) for k in z.files if k.startswith(f"b{b}/")}))
    return z, meta, batches


def _module(meta):
    from gnan_amd import GNAN as standalone
    from gnan_amd import models
    kw = dict(in_channels=meta["F"], out_channels=meta["C"], hidden_channels=meta["H"], bias=True, dropout=0.0, device=DEV)
    v = meta["model"]
    if v.startswith("standalone_tensor"):
        return standalone.TensorGNAN(n_layers=meta["L"], normalize_rho=True, is_graph_task=v.endswith("graph"), **kw)
    if v.startswith("models_tensor"):
        return models.TensorGNAN(n_layers=meta["L"], normalize_rho=True, is_graph_task=v.endswith("graph"),
                                 rho_per_feature=meta["rho_per_feature"], readout_n_layers=0, **kw)
    if v == "models_gnan":
        return models.GNAN(num_layers=meta["L"], normalize_rho=True, rho_per_feature=meta["rho_per_feature"], **kw)
    raise ValueError(v)


@pytest.mark.parametrize("graphed", [False, True])
@pytest.mark.parametrize("name", golden_names("trainer_gnan"))
def test_harness_epochs_on_gnan_modules_match_the_reference_trainer(name, graphed, monkeypatch):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from gnan_amd import harness
    z, meta, batches = _load(name)
    if graphed and meta["optimizer"] == "SGD":
        pytest.skip("torch's SGD has no capturable mode: the harness keeps the eager loop (covered by graphed=False)")
    monkeypatch.setattr(harness, "GRAPHED_STEPS", graphed)
    model = _module(meta)
    model.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0/")}, strict=True)
    model = model.to(DEV).train()
    loss_fn = getattr(torch.nn, meta["loss"])()
    opt = (torch.optim.SGD if meta["optimizer"] == "SGD" else torch.optim.Adam)(model.parameters(), lr=meta["lr"])
    epochs = meta["epochs"]
    hist_tr = z["train_hist"] if epochs > 1 else z["train_ret"][None]
    hist_te = z["test_hist"] if epochs > 1 else z["test_ret"][None]
    for e in range(epochs):
        tr = harness.train_epoch(model, batches, loss_fn, opt, DEV, classify=meta["classify"], compute_auc=False,
                                 is_graph_task=meta["graph"])
        te = harness.test_epoch(model, batches, loss_fn, DEV, classify=meta["classify"], compute_auc=False, val_mask=True,
                                is_graph_task=meta["graph"])
        # Adam's sqrt(v) amplifies float32 round-off of tiny gradients from the second step on; SGD steps are linear in them
        rtol = 2e-4 if meta["optimizer"] == "SGD" else 5e-3
        assert np.allclose(np.array(tr, dtype=np.float64), hist_tr[e], rtol=rtol, atol=1e-5), (e, tr, hist_tr[e])
        assert np.allclose(np.array(te, dtype=np.float64), hist_te[e], rtol=rtol, atol=1e-5), (e, te, hist_te[e])
    assert not model.training                                       # trainer.py:97 leaves eval mode on
    scale = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("sd1/"))
    tol = 2e-5 if meta["optimizer"] == "SGD" else 2e-3
    for k, v in model.state_dict().items():
        assert float(np.abs(v.cpu().numpy() - z["sd1/" + k]).max()) <= tol * scale, k
    if graphed:
        store = harness._steps_of(model)
        if meta["graph"]:
            replays = sum(r["step"].step.graph.replays for r in store.graph.buckets.values() if r["step"] is not None)
        else:
            replays = sum(r.value["step"].graph.replays for r in store.node.entries.values() if r.value["step"] is not None)
        assert replays >= 3, "the captured step was never replayed"
        harness.release_steps(model)


def test_interpretability_exports_come_from_the_kernels_tables():
    """f-4 (notebook cells 4-9): rho on the distinct distances, the shape functions on a value grid and the
    f_k(1) * rho(d) heat-map — exported from the piecewise-linear tables the kernels look up (gnan_pwl_build +
    gnan_fpwl_fwd on the grid), compared with the float64 oracle."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from gnan_amd import interpret
    from gnan_amd.models import TensorGNAN
    torch.manual_seed(0)
    F, C = 15, 3
    m = TensorGNAN(F, C, 3, hidden_channels=64, rho_per_feature=True, device=DEV)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else (2.0 / sum(p.shape)) ** 0.5))
        m.fs[2][0].bias.zero_()                                     # a feature whose kinks all sit at 0 (the grid contains 0)
    m = m.to(DEV).eval()
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    r = interpret.rho_curve(m, 6)
    assert r.shape == (8, C) and r.is_cuda
    assert O.rel_err(r.cpu(), O.rho_lut(sd, 8, dtype=torch.float64)) <= 1e-5
    grid = torch.linspace(-2, 3, 501)
    f = interpret.shape_functions(m, grid)                          # [F, 501, C]
    want = O.feature_mlps(grid.double().view(-1, 1).expand(-1, F).contiguous(), sd).permute(1, 0, 2)
    assert f.shape == (F, 501, C) and O.rel_err(f.cpu(), want) <= 1e-5
    exported = interpret.shape_function_tables(m)                   # the tables themselves: exact breakpoints, values, slopes
    assert exported.off.numel() == F + 1 and exported.val.shape[1] == C
    h = interpret.contribution_heatmap(m, 6)
    f1 = O.feature_mlps(torch.ones(1, F, dtype=torch.float64), sd)[0]              # [F, C]
    assert h.shape == (F, 7, C)
    assert O.rel_err(h.cpu(), f1.unsqueeze(1) * O.rho_lut(sd, 8, dtype=torch.float64)[:7].unsqueeze(0)) <= 1e-5
