"""Pin the oracle: replay every committed golden vector (captured from the imported reference)."""
import numpy as np
import pytest
import torch

from conftest import Golden, golden_names
from helpers import inputs_from, oracle_forward, params_from
from oracle import gnan_oracle as O

MODEL_CASES = [n for n in golden_names() if not any(t in n for t in ("pre_process", "trainer", "run_exp"))]


@pytest.mark.parametrize("name", MODEL_CASES)
def test_forward_fp32_matches_reference(name):
    g = Golden(name)
    y = oracle_forward(g, torch.float32)
    assert tuple(y.shape) == g.out32.shape
    # same ATen ops in the same order: agreement is at the last-ulp level
    assert O.rel_err(y, torch.from_numpy(g.out32)) <= 2e-6


@pytest.mark.parametrize("name", MODEL_CASES)
def test_forward_fp64_matches_reference(name):
    g = Golden(name)
    y = oracle_forward(g, torch.float64)
    assert O.rel_err(y, torch.from_numpy(g.out64)) <= 1e-12


@pytest.mark.parametrize("name", MODEL_CASES)
def test_gradients_fp64_match_reference(name):
    g = Golden(name)
    p = {k: v.clone().requires_grad_(True) for k, v in params_from(g, torch.float64).items()}
    oracle_forward(g, torch.float64, p).pow(2).sum().backward()
    last = "rhos.%d." % (O.n_features(p) - 1)
    for k, ref in g.g64.items():
        # GNAN(rho_per_feature=True): `rho` aliases the modules of `rhos[F-1]` (GNAN.py:108-123,137);
        # named_parameters() reports the shared tensors under the `rhos.{F-1}.` name only.
        src = p["rho." + k[len(last):]] if k.startswith(last) else p[k]
        got = src.grad if src.grad is not None else torch.zeros_like(src)
        scale = max(1.0, float(np.abs(ref).max()))
        assert float((got - torch.from_numpy(ref)).abs().max()) <= 1e-10 * scale, k


@pytest.mark.parametrize("name", golden_names("pre_process"))
def test_pre_process_bit_exact(name):
    g = Golden(name)
    nd, norm = O.pre_process_dense(g.inputs["edge_index"], g.meta["n"])
    assert np.array_equal(nd.numpy(), g.inputs["node_distances"])
    assert np.array_equal(norm.numpy(), g.inputs["normalization_matrix"])


@pytest.mark.parametrize("name", golden_names(("standalone_tensor_node", "models_tensor_node", "models_gnan")))
def test_shell_csr_restatement_matches_dense(name):
    """hop-coded CSR + rest bucket == dense reference, for K in {1, 2, inf}."""
    g = Golden(name)
    m = g.meta
    if m.get("node_ids"):
        pytest.skip("row subset covered by the dense test")
    i = inputs_from(g, torch.float64)
    p = params_from(g, torch.float64)
    pre_rho = m["variant"].startswith("standalone_tensor")
    S = O.feature_mlps(i["x"], p).sum(dim=1)
    hops_full = O.hop_codes_from_dense(i["node_distances"])
    for K in (1, 2, int(hops_full.max())):
        nd_k, norm_k = O.truncate_dense(i["node_distances"], K)
        if pre_rho:
            truth = O.tensor_gnan_forward_standalone(i["x"], nd_k, norm_k, p, m["normalize_rho"], False)
        else:
            truth = O.tensor_gnan_forward_models(i["x"], nd_k, norm_k, p, m["normalize_rho"], False)
        hops = O.hop_codes_from_dense(nd_k)
        D = K + 2
        rowptr, col, code = O.csr_from_hops(hops, K)
        cnt = O.shell_counts(hops, D)
        if not m["normalize_rho"]:
            wtab = O.rho_lut(p, D, torch.float64).unsqueeze(0).expand(len(cnt), -1, -1)
        elif pre_rho:
            wtab = O.row_lut_pre_rho(p, cnt, torch.float64)
        else:
            wtab = O.weight_table(O.rho_lut(p, D, torch.float64), cnt)
        y = O.spmm_csr(rowptr, col, code, S, wtab)
        assert O.rel_err(y, truth) <= 1e-12, (K, O.rel_err(y, truth))
        if not pre_rho:     # the vectorised, differentiable form the full-shape config-3 test uses as float64 truth
            yv = O.spmm_csr_vectorised(rowptr, col, code, S, O.rho_lut(p, D, torch.float64), cnt if m["normalize_rho"] else None)
            assert O.rel_err(yv, truth) <= 1e-12, (K, O.rel_err(yv, truth))
