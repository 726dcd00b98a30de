"""The modules under the reference's OWN loop shape (tests/reference_loop.py restates /root/reference/trainer.py:23-154):
``torch.autograd.set_detect_anomaly(True)`` around the epoch, ``optimizer.zero_grad()``, ``model.forward(data)``, the mask,
the loss, ``loss.backward()``, a stock ``torch.optim`` optimizer built over ``model.parameters()`` and stepped eagerly.
No gnan_amd harness, no captured step, no fused loss.  Numbers: goldens 310-317 — the REFERENCE's trainer driving the
REFERENCE's classes — with the tolerance rule of tests/test_gpu_harness.py."""
import numpy as np
import pytest
import torch

from conftest import golden_names
from test_gpu_harness import DEV, _load, _module

import reference_loop

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_names("trainer_gnan"))
def test_modules_under_the_reference_shaped_loop_in_anomaly_mode(name):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    z, meta, batches = _load(name)
    model = _module(meta)
    model.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0/")}, strict=True)
    model = model.to(DEV).train()
    loss_fn = getattr(torch.nn, meta["loss"])()
    opt = (torch.optim.SGD if meta["optimizer"] == "SGD" else torch.optim.Adam)(model.parameters(), lr=meta["lr"])   # main.py:141
    epochs = meta["epochs"]
    hist_tr = z["train_hist"] if epochs > 1 else z["train_ret"][None]
    hist_te = z["test_hist"] if epochs > 1 else z["test_ret"][None]
    for e in range(epochs):
        assert not torch.is_anomaly_enabled()
        tr = reference_loop.train_epoch(model, batches, loss_fn, opt, DEV, classify=meta["classify"], is_graph_task=meta["graph"])
        te = reference_loop.test_epoch(model, batches, loss_fn, DEV, classify=meta["classify"], val_mask=True,
                                       is_graph_task=meta["graph"])
        # every trainer fixture carries the reference's float64 twin run (tests/golden/make_golden_run.py): the rule of SURVEY
        # section 8c on the trajectory — |loss - loss64| <= max(1e-5, the float32 reference's own gap) of the loss scale;
        # accuracies are hit counts
        for got, h32, h64 in ((tr, hist_tr, z["train_hist64"]), (te, hist_te, z["test_hist64"])):
            scale64 = np.abs(h64[:, 0]).max()
            ref_gap = np.abs(h32[:, 0] - h64[:, 0]).max() / scale64
            err = abs(float(got[0]) - h64[e, 0]) / scale64
            assert err <= max(1e-5, ref_gap), (e, err, ref_gap)
            assert abs(float(got[1]) - h32[e, 1]) <= 1e-6, (e, got, h32[e])
    assert not model.training
    scale = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("sd1/"))
    # parameters after the run: against the float64 run, bounded by the float32 run's own gap
    gap = max(float(np.abs(z["sd1/" + k] - z["sd1_64/" + k]).max()) for k in model.state_dict()) / scale
    for k, v in model.state_dict().items():
        assert float(np.abs(v.cpu().numpy() - z["sd1_64/" + k]).max()) <= max(1e-5, gap) * scale, (k, gap)


@pytest.mark.parametrize("shape", ["node_csr", "node_csr_reference_order", "graph_dense_large", "graph_readout", "graph_readout_large",
                                   "pre_rho"])
def test_every_route_of_the_forward_survives_anomaly_mode(shape):
    """The routes the goldens do not reach (a CSR with a rest bucket and hub rows, the reference aggregation order, a dense
    graph too large for the one-launch kernel, the NAM read-out, the stand-alone file's pre-rho class): forward + backward
    inside anomaly mode must finish, leave finite gradients on every parameter that the plain pass gives one, and give the
    same numbers as outside it."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from gnan_amd import GNAN as standalone
    from gnan_amd import HopGraph, models
    from gnan_amd import synthetic as syn
    from oracle import gnan_oracle as O
    rng = np.random.default_rng(5)
    torch.manual_seed(5)

    class Bag:
        pass
    d = Bag()
    if shape.startswith("node_csr"):
        n, F, C = 20_000, 12, 3
        src, dst = syn.rmat_edges(15, n, 8 * n, seed=0, device=DEV)
        d.x = syn.block_features(n, F, 0, n, seed=1, device=DEV)
        d.edge_index, d.gnan_graph = None, syn.hop1_csr(src, dst, n)
        m = models.TensorGNAN(F, C, 3, hidden_channels=16, rho_per_feature=True, device=DEV)
        if shape.endswith("reference_order"):
            m.aggregation_order = "reference"
    else:
        n = 300 if shape in ("graph_dense_large", "graph_readout_large") else 40
        F, C = 6, 2
        ei = np.stack([rng.integers(0, n, 3 * n), rng.integers(0, n, 3 * n)])
        ei = np.concatenate([ei, ei[::-1]], axis=1)
        nd, norm = O.pre_process_dense(ei, n)
        d.x = torch.cat([torch.from_numpy(rng.random((n, F - 1), dtype=np.float32)), torch.ones(n, 1)], 1).to(DEV)
        d.edge_index, d.node_distances, d.normalization_matrix = torch.from_numpy(ei).to(DEV), nd.to(DEV), norm.to(DEV)
        if shape == "pre_rho":
            m = standalone.TensorGNAN(F, C, 3, hidden_channels=16, is_graph_task=True, device=DEV)
        else:
            m = models.TensorGNAN(F, C, 3, hidden_channels=16, is_graph_task=True, device=DEV,
                                  readout_n_layers=2 if shape.startswith("graph_readout") else 0)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else (2.0 / sum(p.shape)) ** 0.5))
    m = m.to(DEV).eval()

    def once(anomaly):
        m.zero_grad()
        with torch.autograd.set_detect_anomaly(anomaly):
            y = m.forward(d)
            y.pow(2).sum().backward()
        return y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    y0, g0 = once(False)
    if shape == "graph_readout_large":           # the general kernels + NAM read-out (models.py:379-381) against the float64 oracle
        sd64 = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
        truth = O.tensor_gnan_forward_models(d.x.cpu().double(), d.node_distances.cpu().double(),
                                             d.normalization_matrix.cpu().double(), sd64, True, True, 2)
        assert O.rel_err(y0.cpu(), truth) <= 1e-5
    y1, g1 = once(True)
    y2, g2 = once(True)                                                # twice: nothing the first pass left behind trips the next
    assert sorted(g0) == sorted(g1) == sorted(g2) and len(g0) > 0
    assert torch.equal(y0, y1) and torch.equal(y1, y2)
    scale = max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        assert float((g1[k] - g0[k]).abs().max()) <= 1e-6 * scale and float((g2[k] - g0[k]).abs().max()) <= 1e-6 * scale, k
