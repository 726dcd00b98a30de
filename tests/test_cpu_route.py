"""The product's CPU route (``gnan_amd/cpu_route.py``: plain torch on CPU tensors, no kernels, no oracle) against the golden
vectors of the reference, by the rule of SURVEY.md 8c — forward and every parameter gradient.  BASELINE's first configuration
("TensorGNAN on PyTorch CPU") and the reference's default constructor (``device='cpu'``, GNAN.py:10-11) run as written."""
import numpy as np
import pytest
import torch

import gpu_util
from conftest import Golden, golden_names
from helpers import grad_rule, tolerance_ok
from oracle import gnan_oracle as O

MODEL_CASES = [n for n in golden_names() if not any(t in n for t in ("pre_process", "batched", "trainer", "run_exp"))]


@pytest.mark.parametrize("name", MODEL_CASES)
def test_cpu_forward_and_backward_match_golden(name):
    g = Golden(name)
    mod = gpu_util.build_module(g, "cpu")
    data = gpu_util.device_inputs(g, "cpu")
    y = gpu_util.call(mod, g, data)
    assert y.device.type == "cpu" and tuple(y.shape) == g.out32.shape
    ok, e_build, e_ref = tolerance_ok(y.detach(), g.out32, g.out64, floor=1e-5)
    assert ok, f"build err {e_build:.3e} vs fp32-reference err {e_ref:.3e}"
    y.pow(2).sum().backward()
    named = dict(mod.named_parameters())
    ok, e_build, e_ref, where = grad_rule({k: named[k].grad for k in g.g64}, g.g64, g.g32)
    assert ok, f"{where}: build {e_build:.3e} vs fp32-reference {e_ref:.3e}"


def test_default_constructor_runs_on_the_cpu_like_the_reference():
    """``TensorGNAN(in, out, n_layers, hidden)`` with nothing else (device='cpu'), dense inputs, a training step of a stock
    optimizer — and a K = 1 hop-coded CSR with a rest bucket through the same route, against the float64 oracle."""
    from gnan_amd import HopGraph
    from gnan_amd.models import GNAN, TensorGNAN
    rng = np.random.default_rng(0)
    n, f_raw = 60, 4
    ei = np.stack([rng.integers(0, n - 5, 150), rng.integers(0, n - 5, 150)])
    ei = np.concatenate([ei, ei[::-1]], axis=1)
    nd, norm = O.pre_process_dense(ei, n)
    x = torch.cat([torch.from_numpy(rng.random((n, f_raw), dtype=np.float32)), torch.ones(n, 1)], 1)
    m = TensorGNAN(f_raw + 1, 3, 3, hidden_channels=16, rho_per_feature=True)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else (2.0 / sum(p.shape)) ** 0.5))
    data = gpu_util.Bag(x=x, edge_index=torch.from_numpy(ei), node_distances=nd, normalization_matrix=norm)
    sd64 = {k: v.detach().double() for k, v in m.state_dict().items()}
    y = m(data)
    truth = O.tensor_gnan_forward_models(x.double(), nd.double(), norm.double(), sd64, True, False, 0)
    assert O.rel_err(y.detach(), truth) <= 1e-5
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)                       # main.py:141
    before = m.fs[0][0].weight.detach().clone()
    y.pow(2).sum().backward()
    opt.step()
    assert not torch.equal(before, m.fs[0][0].weight.detach())
    # hop-coded CSR, K = 1: listed pairs by code, everything else through the rest bucket
    hops = O.hop_codes_from_dense(nd)
    hops[hops > 1] = -1
    nd1 = torch.from_numpy(np.where(hops >= 0, 1.0 / (hops + 1.0), 0.0).astype(np.float32))
    _, norm1 = O.truncate_dense(nd1, 1)
    rowptr, col, code = O.csr_from_hops(hops, 1)
    g = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(code), n_cols=n, n_codes=3)
    m2 = GNAN(f_raw + 1, 2, num_layers=3, hidden_channels=16)
    with torch.no_grad():
        for _, p in m2.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.4)
    sd2 = {k: v.detach().double() for k, v in m2.state_dict().items()}
    y2 = m2(gpu_util.Bag(x=x, edge_index=None, gnan_graph=g), node_ids=[1, 5, 7])
    truth2 = O.gnan_forward(x.double(), nd1.double(), norm1.double(), sd2, True, [1, 5, 7])
    assert O.rel_err(y2.detach(), truth2) <= 1e-5


def test_cpu_route_is_not_a_fallback():
    """Mixed devices raise; dense inputs that are not of the reference's form raise exactly as on the GPU; and the route
    never imports the oracle."""
    import sys
    from gnan_amd import _lib, cpu_route
    from gnan_amd.models import TensorGNAN
    m = TensorGNAN(3, 1, 2, hidden_channels=4)
    bad = gpu_util.Bag(x=torch.rand(4, 3), node_distances=torch.full((4, 4), 0.3), normalization_matrix=torch.ones(4, 4))
    with pytest.raises(_lib.GnanHipError):
        m(bad)                                              # 0.3 is not 1 / (1 + hop)
    nd = torch.tensor([[1.0, 0.5], [0.5, 1.0]])
    with pytest.raises(_lib.GnanHipError):
        m(gpu_util.Bag(x=torch.rand(2, 3), node_distances=nd, normalization_matrix=torch.full((2, 2), 2.0)))   # counts are 1
    src = open(cpu_route.__file__).read()
    assert "import oracle" not in src and "from oracle" not in src
    meta = torch.empty(2, 3, device="meta")
    with pytest.raises(_lib.GnanHipError):
        cpu_route.applies(m, meta)                          # module on the CPU, input elsewhere


def test_run_loop_on_the_cpu():
    """The build's counterpart of main.py (``gnan_amd.run.run_exp``) end to end on the CPU — BASELINE config 1's plumbing at a
    small shape: epochs run, losses are finite, the reference's checkpoint files appear, and the trained weights equal the same
    loop over torch's own per-layer ``parameters()`` (the flat buffers ``run_exp`` optimises are the same numbers)."""
    import tempfile
    from gnan_amd import run
    rng = np.random.default_rng(3)
    n, f_raw, C = 90, 6, 3
    ei = np.stack([rng.integers(0, n, 200), rng.integers(0, n, 200)])
    ei = np.concatenate([ei, ei[::-1]], axis=1)
    nd, norm = O.pre_process_dense(ei, n)
    x = torch.cat([torch.from_numpy(rng.random((n, f_raw), dtype=np.float32)), torch.ones(n, 1)], 1)

    class Data(gpu_util.Bag):
        def to(self, device):
            return self
    mask = torch.zeros(n, dtype=torch.bool)
    mask[::2] = True
    data = Data(x=x, edge_index=torch.from_numpy(ei), node_distances=nd, normalization_matrix=norm,
                y=torch.from_numpy(rng.integers(0, C, n)), train_mask=mask, val_mask=~mask, test_mask=~mask)
    with tempfile.TemporaryDirectory() as tmp:
        torch.manual_seed(0)
        res = run.run_exp([data], [data], [data], f_raw + 1, [0], 2, True, 0.0, "gnan", 3, False, 0.0, 8, 1e-2, 1e-9, "toy", "t", False,
                          True, False, C, C, device=torch.device("cpu"), checkpoint_dir=tmp, log=lambda *_: None)[0]
        import os
        assert len(res["epochs"]) == 3 and all(np.isfinite(e["train_loss"]) for e in res["epochs"])
        assert any(name.endswith("_best_train_loss.pt") for _, name in res["checkpoints"]) and os.listdir(tmp)
    assert res["epochs"][-1]["train_loss"] < res["epochs"][0]["train_loss"]
    # the same three epochs by hand over torch's own parameters()
    from gnan_amd.models import GNAN
    torch.manual_seed(0)
    twin = GNAN(in_channels=f_raw + 1, hidden_channels=8, num_layers=2, out_channels=C, dropout=0.0)
    opt = torch.optim.Adam(twin.parameters(), lr=1e-2, weight_decay=0.0)
    loss_fn = torch.nn.CrossEntropyLoss()
    for _ in range(3):
        opt.zero_grad()
        loss_fn(twin(data)[mask], data.y[mask]).backward()
        opt.step()
    for (k, p), (_, q) in zip(res["model"].named_parameters(), twin.named_parameters()):
        assert float((p - q).abs().max()) <= 1e-6 * max(1e-3, float(q.abs().max())), k
