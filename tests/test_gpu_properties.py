"""Property tests on the GPU (hypothesis): the three ways of feeding a graph — the reference's dense matrices,
edge_index through the GPU BFS, a full-depth hop-coded CSR — give the same model outputs, and they match the oracle,
for random small graphs incl. directed ones, isolated nodes, several components, every depth / bias / normalisation
/ rho-width combination."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import gnan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


@st.composite
def problems(draw):
    n = draw(st.integers(3, 40))
    m = draw(st.integers(0, 3 * n))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    ei = rng.integers(0, n, (2, m))
    ei = ei[:, ei[0] != ei[1]]
    ei = np.unique(ei, axis=1)
    if not draw(st.booleans()) and ei.shape[1]:
        ei = np.unique(np.concatenate([ei, ei[::-1]], axis=1), axis=1)          # undirected
    return dict(n=n, ei=ei, seed=seed, L=draw(st.integers(1, 3)), bias=draw(st.booleans()),
                normalize=draw(st.booleans()), per_feature=draw(st.booleans()), C=draw(st.integers(1, 4)),
                F=draw(st.integers(1, 5)), variant=draw(st.sampled_from(["models_tensor", "standalone_tensor", "gnan"])))


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(problems())
def test_dense_bfs_csr_and_oracle_agree(pr):
    import gnan_amd  # noqa: F401
    from gnan_amd import GNAN as standalone
    from gnan_amd import HopGraph, models
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    n, ei = pr["n"], pr["ei"]
    nd, norm = O.pre_process_dense(ei, n) if ei.shape[1] else (torch.eye(n), None)
    if norm is None:
        norm = torch.where(torch.eye(n) > 0, torch.ones(n, n), torch.full((n, n), float(n - 1)))
    rng = np.random.default_rng(pr["seed"])
    x = torch.from_numpy(rng.random((n, pr["F"]), dtype=np.float32))
    kw = dict(in_channels=pr["F"], out_channels=pr["C"], hidden_channels=8, bias=pr["bias"],
              normalize_rho=pr["normalize"], device=DEV)
    if pr["variant"] == "models_tensor":
        mod = models.TensorGNAN(n_layers=pr["L"], rho_per_feature=pr["per_feature"], **kw)
        ref = lambda xx, a, b, p: O.tensor_gnan_forward_models(xx, a, b, p, pr["normalize"], False)
    elif pr["variant"] == "standalone_tensor":
        mod = standalone.TensorGNAN(n_layers=pr["L"], **kw)
        ref = lambda xx, a, b, p: O.tensor_gnan_forward_standalone(xx, a, b, p, pr["normalize"], False)
    else:
        mod = models.GNAN(num_layers=pr["L"], rho_per_feature=pr["per_feature"], **kw)
        ref = lambda xx, a, b, p: O.gnan_forward(xx, a, b, p, pr["normalize"])
    gen = torch.Generator().manual_seed(pr["seed"] % 1000)
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else (2.0 / sum(p.shape)) ** 0.5))
    p64 = {k: v.detach().double() for k, v in mod.state_dict().items()}
    truth = ref(x.double(), nd.double(), norm.double(), p64)
    # within max(1e-5, the float32 reference's own distance from the float64 truth)
    floor = max(1e-5, O.rel_err(ref(x, nd, norm, {k: v.float() for k, v in p64.items()}), truth))
    mod = mod.to(DEV).eval()
    xd = x.to(DEV)
    hops = O.hop_codes_from_dense(nd)
    K = max(int(hops.max()), 0)
    rowptr, col, code = O.csr_from_hops(hops, K)
    feeds = {
        "dense": Bag(x=xd, edge_index=None, node_distances=nd.to(DEV), normalization_matrix=norm.to(DEV)),
        "csr": Bag(x=xd, edge_index=None, gnan_graph=HopGraph.from_csr(
            torch.from_numpy(rowptr).to(DEV), torch.from_numpy(col).to(DEV), torch.from_numpy(code).to(DEV),
            n_cols=n, n_codes=K + 2)),
    }
    if ei.shape[1]:
        feeds["bfs"] = Bag(x=xd, edge_index=None, gnan_graph=HopGraph.from_edge_index(torch.from_numpy(ei).to(DEV), n))
    outs = {}
    with torch.no_grad():
        for name, d in feeds.items():
            outs[name] = mod.forward(d).cpu()
            assert O.rel_err(outs[name], truth) <= floor, (name, O.rel_err(outs[name], truth), floor)
    if "bfs" in outs:
        assert torch.equal(outs["bfs"], outs["dense"])          # identical codes and counts -> identical arithmetic
