"""Kernel-level GPU tests: each C-ABI entry point against the oracle on seeded inputs, incl. edge cases
(empty rows, hub rows, ragged widths, row subsets) and size-independent properties (linearity)."""
import numpy as np
import pytest
import torch

from oracle import gnan_oracle as O
from helpers import TWO_FLOORS, assert_grads_rule, assert_rule
from gnan_amd import aggregate  # noqa: E402  (the aggregation's switches are patched below)

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")


def _mlp_state(F, L, H, C, bias, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in range(F):
        dims = [1] + [H] * (L - 1) + [C]
        for li in range(L):
            sd[f"fs.{k}.{3 * li}.weight"] = torch.randn(dims[li + 1], dims[li], generator=g) * (2.0 / (dims[li] + dims[li + 1])) ** 0.5
            if bias:
                sd[f"fs.{k}.{3 * li}.bias"] = torch.randn(dims[li + 1], generator=g) * 0.5
    return sd


def _stack(sd, F, L, H, C, bias):
    from gnan_amd.functional import StackedMLP

    def cat(li, what):
        return torch.stack([sd[f"fs.{k}.{3 * li}.{what}"] for k in range(F)], 0).to(DEV)

    if L == 1:
        return StackedMLP(None, None, None, None, cat(0, "weight")[..., 0], cat(0, "bias") if bias else None, 1, 0, C, F)
    w_mid = b_mid = None
    if L > 2:
        w_mid = torch.stack([cat(li, "weight") for li in range(1, L - 1)], 0)
        b_mid = torch.stack([cat(li, "bias") for li in range(1, L - 1)], 0) if bias else None
    return StackedMLP(cat(0, "weight")[..., 0], cat(0, "bias") if bias else None, w_mid, b_mid,
                      cat(L - 1, "weight"), cat(L - 1, "bias") if bias else None, L, H, C, F)


@pytest.mark.parametrize("F,L,H,C,bias", [
    (3, 1, 0, 2, True), (4, 2, 8, 3, True), (5, 3, 8, 1, True), (9, 3, 32, 5, False),
    (15, 3, 64, 1, True), (7, 4, 16, 7, True), (3, 3, 20, 40, True), (64, 3, 64, 1, True),
    (6, 3, 33, 2, True), (5, 4, 64, 8, False), (2, 5, 16, 3, True), (3, 3, 96, 2, True), (1, 3, 64, 4, True),
    (129, 3, 32, 1, True), (33, 3, 16, 1, False),                        # ragged feature counts: partial last group
])
@pytest.mark.parametrize("sum_features", [True, False])
@pytest.mark.parametrize("algo", ["lane", "auto", "pwl"])
def test_feature_mlps_vs_oracle(F, L, H, C, bias, sum_features, algo, monkeypatch):
    """All three strategies: lane-per-node kernel, the matrix-core kernel AUTO picks for 3<=L<=4, H<=64,
    C<=8 at this size, and the exact piecewise-linear table look-up AUTO picks for large batches."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO",
                        {"lane": _lib.FMLP_LANE, "auto": _lib.FMLP_AUTO, "pwl": _lib.FMLP_PWL}[algo])
    n = 203
    sd = _mlp_state(F, L, max(H, 1), C, bias, seed=F * 100 + L)
    x = torch.rand(n, F, generator=torch.Generator().manual_seed(1))
    truth = O.feature_mlps(x.double(), {k: v.double() for k, v in sd.items()})
    ref32 = O.feature_mlps(x, sd)
    if sum_features:
        truth, ref32 = truth.sum(1), ref32.sum(1)
    else:
        truth, ref32 = truth.reshape(n, -1), ref32.reshape(n, -1)
    with torch.no_grad():
        y = feature_mlps(x.to(DEV), _stack(sd, F, L, H, C, bias), sum_features).cpu()
    assert y.shape == truth.shape
    e_build, e_ref = O.rel_err(y, truth), O.rel_err(ref32, truth)
    assert e_build <= max(1e-5, e_ref), (e_build, e_ref)


@pytest.mark.parametrize("n,F,L,H,C,bias,sum_features", [
    (3, 1, 3, 64, 1, True, False), (3, 1, 3, 64, 1, False, True), (64, 1, 3, 64, 7, True, False), (5, 1, 2, 64, 3, True, False),
    (4, 1, 3, 33, 2, True, False), (2, 16, 3, 16, 4, False, False), (1, 64, 3, 64, 1, True, False), (7, 1, 3, 8, 40, True, True),
])
def test_a_handful_of_evaluations_take_the_point_kernel(n, F, L, H, C, bias, sum_features):
    """n * F <= 64 evaluations under AUTO (rho on the D hop codes of a truncated graph, on every training step): one wave per
    (node, feature), lane = hidden unit (fmlp_point_kernel) — against the float64 oracle and against the lane kernel, with
    gradients through the same route (gnan_fmlp_bwd)."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    sd = _mlp_state(F, L, H, C, bias, seed=n * 10 + F)
    x = (torch.rand(n, F, generator=torch.Generator().manual_seed(n)) * 2 - 0.5)
    truth = O.feature_mlps(x.double(), {k: v.double() for k, v in sd.items()})
    ref32 = O.feature_mlps(x, sd)
    truth, ref32 = (truth.sum(1), ref32.sum(1)) if sum_features else (truth.reshape(n, -1), ref32.reshape(n, -1))
    st = _stack(sd, F, L, H, C, bias)
    assert _lib.lib().gnan_fmlp_fwd_workspace_bytes is not None
    leaves = [None if t is None else t.to(DEV).requires_grad_(True) for t in st[:6]]
    y = feature_mlps(x.to(DEV), functional.StackedMLP(*leaves, *st[6:]), sum_features)
    e_build, e_ref = O.rel_err(y.detach().cpu(), truth), O.rel_err(ref32, truth)
    assert e_build <= max(1e-5, e_ref), (e_build, e_ref)
    y.square().sum().backward()
    functional.FMLP_ALGO = _lib.FMLP_LANE
    try:
        leaves2 = [None if t is None else t.detach().clone().requires_grad_(True) for t in leaves]
        y2 = feature_mlps(x.to(DEV), functional.StackedMLP(*leaves2, *st[6:]), sum_features)
        y2.square().sum().backward()
    finally:
        functional.FMLP_ALGO = _lib.FMLP_AUTO
    assert float((y - y2).abs().max()) <= 1e-5 * max(1e-30, float(y2.abs().max()))
    for a, b in zip(leaves, leaves2):
        if a is not None:
            assert float((a.grad - b.grad).abs().max()) <= TWO_FLOORS * max(1e-30, float(b.grad.abs().max()))      # two routes


def _random_csr(n_rows, n_cols, K, rng, hubs=()):
    deg = rng.poisson(6, n_rows)
    deg[rng.random(n_rows) < 0.1] = 0                    # empty rows
    for r, d in hubs:
        deg[r] = d
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    col = rng.integers(0, n_cols, nnz).astype(np.int32)
    code = rng.integers(0, K + 1, nnz).astype(np.uint8)
    return rowptr, col, code


def _graph(rowptr, col, code, n_cols, D, idx_dtype=torch.int64):
    from gnan_amd import HopGraph
    return HopGraph.from_csr(torch.from_numpy(rowptr).to(idx_dtype).to(DEV), torch.from_numpy(col).to(DEV),
                             torch.from_numpy(code).to(DEV), n_cols=n_cols, n_codes=D)


def _cnt_np(rowptr, code, n_cols, D):
    n = len(rowptr) - 1
    cnt = np.zeros((n, D), dtype=np.int64)
    for i in range(n):
        seg = code[rowptr[i]:rowptr[i + 1]]
        cnt[i, :D - 1] = np.bincount(seg, minlength=D - 1)[:D - 1]
        cnt[i, D - 1] = n_cols - len(seg)
    return cnt


@pytest.mark.parametrize("W,Cw", [(1, 1), (3, 1), (3, 3), (7, 7), (8, 1), (40, 40), (40, 8), (64, 1), (100, 1), (300, 1)])
@pytest.mark.parametrize("K,use_cnt,per_row", [(1, True, False), (2, False, False), (5, True, False), (1, False, True)])
def test_spmm_csr_vs_oracle(W, Cw, K, use_cnt, per_row):
    from gnan_amd.aggregate import spmm_launch
    rng = np.random.default_rng(W * 10 + K)
    n_rows, n_cols, D = 301, 257, K + 2
    rowptr, col, code = _random_csr(n_rows, n_cols, K, rng, hubs=[(5, 700), (17, 2500)])
    cnt = _cnt_np(rowptr, code, n_cols, D)
    S = torch.from_numpy(rng.standard_normal((n_cols, W)).astype(np.float32))
    lut = torch.from_numpy(rng.standard_normal((n_rows, D, Cw) if per_row else (D, Cw)).astype(np.float32))
    wt64 = (lut.double() if per_row else O.weight_table(lut.double(), cnt if use_cnt else None)
            .expand(n_rows, -1, -1))
    truth = O.spmm_csr(rowptr, col, code, S.double(), wt64)
    g = _graph(rowptr, col, code, n_cols, D, torch.int32 if W % 2 else torch.int64)
    assert np.array_equal(g.cnt.cpu().numpy(), cnt)
    y = spmm_launch(g, S.to(DEV), lut.to(DEV), use_cnt, with_rest=True).cpu()
    assert O.rel_err(y, truth) <= 1e-5, O.rel_err(y, truth)


def test_spmm_row_subset_and_dense_equals_csr():
    from gnan_amd import HopGraph
    from gnan_amd.aggregate import spmm_launch
    rng = np.random.default_rng(7)
    n, D, W = 90, 6, 5
    hops = rng.integers(-1, D - 1, (n, n)).astype(np.int32)          # -1 = unreachable
    nd = torch.zeros(n, n)
    nd[torch.from_numpy(hops >= 0)] = 1.0 / (torch.from_numpy(hops[hops >= 0]).float() + 1.0)
    dense = HopGraph.from_dense(nd.to(DEV))
    rowptr, col, code = O.csr_from_hops(hops, D - 2)
    csr = _graph(rowptr, col, code, n, dense.n_codes)
    assert np.array_equal(csr.cnt.cpu().numpy(), dense.cnt.cpu().numpy())
    S = torch.randn(n, W, generator=torch.Generator().manual_seed(3)).to(DEV)
    lut = torch.randn(dense.n_codes, 1, generator=torch.Generator().manual_seed(4)).to(DEV)
    y_dense = spmm_launch(dense, S, lut, True, with_rest=False)
    y_csr = spmm_launch(csr, S, lut, True, with_rest=True)
    assert O.rel_err(y_csr.cpu(), y_dense.cpu().double()) <= 2e-6
    ids = torch.tensor([4, 0, 77, 4], dtype=torch.int32, device=DEV)
    y_sub = spmm_launch(dense, S, lut, True, with_rest=False, row_ids=ids)
    assert torch.equal(y_sub, y_dense[ids.long()])                       # same arithmetic, bit-identical
    y_sub = spmm_launch(csr, S, lut, True, with_rest=True, row_ids=ids)
    assert torch.equal(y_sub, y_csr[ids.long()])


@pytest.mark.parametrize("n,W,reduce_cr,Cw", [(700, 7, 0, 1), (2500, 64, 1, 1), (1030, 6, 2, 2), (600, 3, 0, 3)])
def test_dense_rows_sliced_over_workgroups(n, W, reduce_cr, Cw, monkeypatch):
    """Cora-sized dense inputs: every row cut into column slices (dense_slice_plan) == one lane group per row (the
    arithmetic differs only in summation order) == the oracle; row subsets, transposed use and gradients included."""
    from gnan_amd import HopGraph, functional
    from gnan_amd.aggregate import rho_aggregate, spmm_launch
    rng = np.random.default_rng(n + W)
    D = 7
    hops = rng.integers(-1, D - 1, (n, n)).astype(np.int32)
    nd = torch.zeros(n, n)
    nd[torch.from_numpy(hops >= 0)] = 1.0 / (torch.from_numpy(hops[hops >= 0]).float() + 1.0)
    g = HopGraph.from_dense(nd.to(DEV))
    S = torch.randn(n, W, generator=torch.Generator().manual_seed(3)).to(DEV)
    lut = torch.randn(g.n_codes, Cw, generator=torch.Generator().manual_seed(4)).to(DEV)
    ids = torch.from_numpy(rng.integers(0, n, 97).astype(np.int32)).to(DEV)
    sliced = spmm_launch(g, S, lut, True, with_rest=False, reduce_cr=reduce_cr)
    sliced_sub = spmm_launch(g, S, lut, True, with_rest=False, row_ids=ids, reduce_cr=reduce_cr)
    sliced_t = spmm_launch(g.transposed(), S, lut, True, False, None, weight_by_col=True)
    monkeypatch.setattr(aggregate, "DENSE_SLICE_MAX_ROWS", 0)             # row blocks only
    plain = spmm_launch(g, S, lut, True, with_rest=False, reduce_cr=reduce_cr)
    plain_t = spmm_launch(g.transposed(), S, lut, True, False, None, weight_by_col=True)
    assert float((sliced - plain).abs().max()) <= TWO_FLOORS * float(plain.abs().max())              # two routes; the truth: below
    assert float((sliced_t - plain_t).abs().max()) <= TWO_FLOORS * float(plain_t.abs().max())
    assert torch.equal(sliced_sub, sliced[ids.long()])
    rowptr, col, code = O.csr_from_hops(hops, D - 2)
    wt = O.weight_table(lut.cpu().double(), g.cnt.cpu().numpy())
    truth = O.spmm_csr(rowptr, col, code, S.cpu().double(), wt)           # rest bucket: unreachable pairs, weight lut[D-1]
    if reduce_cr:
        truth = truth.view(n, -1, reduce_cr).sum(1)
    assert O.rel_err(sliced.cpu(), truth) <= 1e-5
    monkeypatch.undo()
    Sg, lg = S.clone().requires_grad_(True), lut.clone().requires_grad_(True)
    rho_aggregate(g, Sg, lg, True).pow(2).sum().backward()
    monkeypatch.setattr(aggregate, "DENSE_SLICE_MAX_ROWS", 0)
    Sp, lp = S.clone().requires_grad_(True), lut.clone().requires_grad_(True)
    rho_aggregate(g, Sp, lp, True).pow(2).sum().backward()
    assert float((Sg.grad - Sp.grad).abs().max()) <= TWO_FLOORS * float(Sp.grad.abs().max())         # two routes ...
    assert float((lg.grad - lp.grad).abs().max()) <= TWO_FLOORS * float(lp.grad.abs().max())
    # ... and both against float64 autograd through the oracle, by the rule
    def oracle_grads_at(dtype):
        S_, l_ = S.cpu().to(dtype).requires_grad_(True), lut.cpu().to(dtype).requires_grad_(True)
        O.spmm_csr(rowptr, col, code, S_, O.weight_table(l_, g.cnt.cpu().numpy())).pow(2).sum().backward()
        return [S_.grad, l_.grad]
    t64 = oracle_grads_at(torch.float64)
    for grads in ([Sg.grad, lg.grad], [Sp.grad, lp.grad]):
        assert_grads_rule(grads, t64, lambda: oracle_grads_at(torch.float32), "dense rows, sliced / unsliced")


def test_spmm_is_linear_in_the_operand_at_scale():
    """Size-independent property at a size the oracle cannot reach: A(aS1 + S2) == a A S1 + A S2."""
    from gnan_amd.aggregate import spmm_launch
    rng = np.random.default_rng(11)
    n, K, W = 200_000, 1, 64
    deg = np.minimum(rng.zipf(1.8, n), 50_000)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    g = _graph(rowptr, rng.integers(0, n, nnz).astype(np.int32), rng.integers(0, K + 1, nnz).astype(np.uint8), n, K + 2)
    S1 = torch.randn(n, W, device=DEV)
    S2 = torch.randn(n, W, device=DEV)
    lut = torch.tensor([[1.0], [0.5], [0.01]], device=DEV)
    y = spmm_launch(g, 2.0 * S1 + S2, lut, True, True)
    y12 = 2.0 * spmm_launch(g, S1, lut, True, True) + spmm_launch(g, S2, lut, True, True)
    # three results of the kernel, each within the floor of its truth: |A(2 S1 + S2) - (2 A S1 + A S2)| <= 1e-5 (|y| + 2 |A S1| + |A S2|)
    a1, a2 = spmm_launch(g, S1, lut, True, True), spmm_launch(g, S2, lut, True, True)
    assert float((y - y12).abs().max()) <= 1e-5 * float(y.abs().max() + 2 * a1.abs().max() + a2.abs().max())
    # deterministic: the hub-row slices are reduced in a fixed order
    assert torch.equal(y, spmm_launch(g, 2.0 * S1 + S2, lut, True, True))


def test_empty_inputs():
    from gnan_amd.aggregate import spmm_launch
    g = _graph(np.zeros(1, dtype=np.int64), np.zeros(0, np.int32), np.zeros(0, np.uint8), 5, 3)
    y = spmm_launch(g, torch.randn(5, 4, device=DEV), torch.randn(3, 1, device=DEV), True, True)
    assert y.shape == (0, 4)


@pytest.mark.parametrize("per_row,use_cnt,Cw", [(False, True, 1), (False, False, 3), (True, False, 3)])
def test_rho_aggregate_gradients_vs_autograd_oracle(per_row, use_cnt, Cw):
    """Backward kernels (transposed SpMM + shell sums) vs torch autograd through the oracle, in float64."""
    from gnan_amd.aggregate import rho_aggregate
    rng = np.random.default_rng(5)
    n, K, W = 120, 2, 6
    D = K + 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(3, 900)])
    cnt = _cnt_np(rowptr, code, n, D)
    S = torch.from_numpy(rng.standard_normal((n, W)))
    lut = torch.from_numpy(rng.standard_normal((n, D, Cw) if per_row else (D, Cw)))
    S64, lut64 = S.clone().requires_grad_(True), lut.clone().requires_grad_(True)
    wt = lut64 if per_row else O.weight_table(lut64, cnt if use_cnt else None).expand(n, -1, -1)
    O.spmm_csr(rowptr, col, code, S64, wt).pow(2).sum().backward()
    g = _graph(rowptr, col, code, n, D)
    Sd = S.float().to(DEV).requires_grad_(True)
    lutd = lut.float().to(DEV).requires_grad_(True)
    rho_aggregate(g, Sd, lutd, use_cnt).pow(2).sum().backward()
    def oracle32():
        S32, l32 = S.float().requires_grad_(True), lut.float().requires_grad_(True)
        w32 = l32 if per_row else O.weight_table(l32, cnt if use_cnt else None).expand(n, -1, -1)
        O.spmm_csr(rowptr, col, code, S32, w32).pow(2).sum().backward()
        return S32.grad, l32.grad
    assert_rule(Sd.grad, S64.grad, lambda: oracle32()[0], "dS")                     # each gradient against its own largest entry
    assert_rule(lutd.grad, lut64.grad, lambda: oracle32()[1], "dlut")


def test_matrix_core_kernel_is_the_one_auto_picks_and_rejects_foreign_shapes(monkeypatch):
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_MFMA)
    sd = _mlp_state(2, 3, 96, 1, True, seed=0)                  # H = 96 > 64: outside the MFMA kernel
    x = torch.rand(10, 2, device=DEV)
    with pytest.raises(_lib.GnanHipError, match="MFMA path covers"):
        feature_mlps(x, _stack(sd, 2, 3, 96, 1, True), True)
    sd = _mlp_state(2, 3, 64, 1, True, seed=0)
    y_mfma = feature_mlps(x, _stack(sd, 2, 3, 64, 1, True), True)
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_LANE)
    y_lane = feature_mlps(x, _stack(sd, 2, 3, 64, 1, True), True)
    assert O.rel_err(y_mfma.cpu(), y_lane.cpu().double()) <= 2e-6


@pytest.mark.parametrize("algo", ["mfma", "pwl"])
def test_feature_mlps_large_ragged_batch(algo, monkeypatch):
    """Property check at a size the oracle would take minutes for: two disjoint halves == the whole."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_MFMA if algo == "mfma" else _lib.FMLP_PWL)
    F, L, H, C = 64, 3, 64, 1
    sd = _mlp_state(F, L, H, C, True, seed=9)
    st = _stack(sd, F, L, H, C, True)
    x = torch.rand(100_003, F, device=DEV)
    whole = feature_mlps(x, st, False)
    assert torch.equal(whole[:50_001], feature_mlps(x[:50_001], st, False))
    assert torch.equal(whole[50_001:], feature_mlps(x[50_001:], st, False))
    sub = O.feature_mlps(x[:97].cpu().double(), {k: v.double() for k, v in sd.items()}).reshape(97, -1)
    assert O.rel_err(whole[:97].cpu(), sub) <= 1e-5


def test_table_lookup_matches_matrix_core_kernel_at_scale(monkeypatch):
    """1M nodes x 64 features: the two independent evaluation strategies agree to fp32 round-off."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    F, L, H, C = 64, 3, 64, 1
    sd = _mlp_state(F, L, H, C, True, seed=4)
    st = _stack(sd, F, L, H, C, True)
    x = torch.rand(1_000_000, F, device=DEV) * 3 - 1
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_MFMA)
    a = feature_mlps(x, st, False)
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    b = feature_mlps(x, st, False)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    sa, sb = feature_mlps(x, st, True), None
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_MFMA)
    sb = feature_mlps(x, st, True)
    assert float((sa - sb).abs().max()) <= 1e-5 * float(sb.abs().max())


@pytest.mark.parametrize("n,W", [(1, 1), (1000, 7), (4097, 64), (300_000, 64), (50_000, 3), (169_343, 1), (8191, 4), (8193, 2),
                                 (2_000_000, 1), (262_145, 4), (65_539, 1), (1_000_001, 1)])
def test_column_sums(n, W):
    from gnan_amd.functional import column_sums
    S = torch.randn(n, W + 1, generator=torch.Generator().manual_seed(n)).to(DEV)[:, :W]      # (strided rows)
    for S in (S, S.contiguous()):
        got = column_sums(S).cpu().double()
        want = S.cpu().double().sum(0)
        assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(S.abs().sum(0).max()))
        assert torch.equal(column_sums(S), column_sums(S))          # fixed reduction order


@pytest.mark.parametrize("W,cr", [(1, 0), (1, 1), (8, 0), (8, 2), (24, 0), (24, 4), (64, 0), (64, 1), (320, 0), (320, 4)])
def test_hub_rows_with_many_slices(W, cr):
    """Hub rows of 20 and 41 slices: the fix-up kernel's eight-loads-at-a-time loop, its slice lanes for operands narrower
    than a wave (W = 1, 8, 24) and its column passes for wider ones (W = 320), with and without the fused feature sum."""
    from gnan_amd.aggregate import spmm_launch
    rng = np.random.default_rng(1000 + W * 3 + cr)
    n_rows, n_cols, K = 64, 5000, 1
    D = K + 2
    rowptr, col, code = _random_csr(n_rows, n_cols, K, rng, hubs=[(3, 40000), (40, 83001), (41, 513)])
    cnt = _cnt_np(rowptr, code, n_cols, D)
    S = torch.from_numpy(rng.standard_normal((n_cols, W)).astype(np.float32))
    lut = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32))
    wt64 = O.weight_table(lut.double(), cnt).expand(n_rows, -1, -1)
    truth = O.spmm_csr(rowptr, col, code, S.double(), wt64)
    g = _graph(rowptr, col, code, n_cols, D)
    y = spmm_launch(g, S.to(DEV), lut.to(DEV), True, with_rest=True, reduce_cr=cr).cpu()
    if cr:
        truth = truth.view(n_rows, W // cr, cr).sum(1)
    assert O.rel_err(y, truth) <= 1e-5, O.rel_err(y, truth)
    again = spmm_launch(g, S.to(DEV), lut.to(DEV), True, with_rest=True, reduce_cr=cr).cpu()
    assert torch.equal(y, again)                                     # fixed reduction order


@pytest.mark.parametrize("W,Cw,cr,per_row,use_cnt", [(64, 1, 1, False, True), (64, 1, 0, False, True), (8, 4, 4, False, True),
                                                      (12, 4, 2, False, False), (6, 3, 0, True, False), (64, 1, 4, False, True)])
def test_rest_bucket_total_can_be_added_afterwards(W, Cw, cr, per_row, use_cnt):
    """aggregate(total) == aggregate(zero sums) + rest_total_term(total): what lets a multi-rank forward overlap the
    all-reduce of the column sums with the aggregation (rows, hub slices, fused read-out, per-row tables)."""
    from gnan_amd.aggregate import rest_total_term, spmm_launch
    rng = np.random.default_rng(500 + W + Cw + cr)
    n, K = 700, 1
    D = K + 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(11, 900), (300, 5000)])
    g = _graph(rowptr, col, code, n, D)
    S = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    lut = torch.from_numpy(rng.standard_normal((n, D, Cw) if per_row else (D, Cw)).astype(np.float32)).to(DEV)
    total = torch.from_numpy(rng.standard_normal(W).astype(np.float32)).to(DEV) * 50       # NOT the sums of S: any vector
    fused = spmm_launch(g, S, lut, use_cnt, True, s_total=total, reduce_cr=cr)
    split = spmm_launch(g, S, lut, use_cnt, True, s_total=torch.zeros_like(total), reduce_cr=cr)
    split = split + rest_total_term(g, lut, use_cnt, total, cr)
    assert split.shape == fused.shape
    assert O.rel_err(split.cpu(), fused.double().cpu()) <= 2e-6


@pytest.mark.parametrize("W,cr", [(64, 1), (64, 4), (8, 2), (6, 2), (6, 1), (300, 4), (3, 1), (1, 1)])
def test_fused_feature_sum_equals_unfused(W, cr):
    """reduce_cr: per-channel sums over the operand columns in the kernel epilogue (rows, hub slices, dense)."""
    from gnan_amd import HopGraph
    from gnan_amd.aggregate import spmm_launch
    rng = np.random.default_rng(W * 7 + cr)
    n, K = 400, 2
    D = K + 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(9, 800), (100, 3000)])
    g = _graph(rowptr, col, code, n, D)
    S = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    lut = torch.from_numpy(rng.standard_normal((D, cr)).astype(np.float32)).to(DEV)
    full = spmm_launch(g, S, lut, True, True)
    want = full.double().view(n, W // cr, cr).sum(1)
    got = spmm_launch(g, S, lut, True, True, reduce_cr=cr)
    assert got.shape == (n, cr)
    assert O.rel_err(got.cpu(), want.cpu()) <= 2e-6
    nd = torch.zeros(n, n)
    hops = rng.integers(-1, 3, (n, n))
    nd[torch.from_numpy(hops >= 0)] = 1.0 / (torch.from_numpy(hops[hops >= 0]).float() + 1.0)
    dense = HopGraph.from_dense(nd.to(DEV))
    lut_d = torch.from_numpy(rng.standard_normal((dense.n_codes, cr)).astype(np.float32)).to(DEV)
    full = spmm_launch(dense, S, lut_d, True, False)
    got = spmm_launch(dense, S, lut_d, True, False, reduce_cr=cr)
    assert O.rel_err(got.cpu(), full.double().view(n, W // cr, cr).sum(1).cpu()) <= 2e-6


def test_fused_feature_sum_gradients():
    from gnan_amd.aggregate import rho_aggregate
    rng = np.random.default_rng(3)
    n, K, F, C = 150, 1, 5, 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(2, 700)])
    g = _graph(rowptr, col, code, n, K + 2)
    S0 = torch.from_numpy(rng.standard_normal((n, F * C)).astype(np.float32)).to(DEV)
    l0 = torch.from_numpy(rng.standard_normal((K + 2, C)).astype(np.float32)).to(DEV)
    grads = []
    for fused in (C, 0):
        S, lut = S0.clone().requires_grad_(True), l0.clone().requires_grad_(True)
        Y = rho_aggregate(g, S, lut, True, reduce_channels=fused)
        if not fused:
            Y = Y.view(n, F, C).sum(1)
        Y.pow(2).sum().backward()
        grads.append((S.grad.clone(), lut.grad.clone()))
    assert O.rel_err(grads[0][0].cpu(), grads[1][0].cpu().double()) <= 1e-5
    assert O.rel_err(grads[0][1].cpu(), grads[1][1].cpu().double()) <= 1e-5


@pytest.mark.parametrize("F,L,H,C,sum_features", [(5, 3, 8, 1, True), (20, 3, 16, 2, False), (64, 3, 64, 1, False),
                                                   (3, 3, 16, 40, True), (64, 3, 64, 1, True), (32, 2, 20, 1, True),
                                                   (8, 3, 16, 1, False), (48, 3, 64, 1, False),
                                                   (129, 3, 32, 1, True), (33, 3, 16, 1, False), (17, 2, 8, 1, True)])
@pytest.mark.parametrize("fixed", [True, False])
def test_pwl_moments_kernel_vs_reference(F, L, H, C, sum_features, fixed, monkeypatch):
    from gnan_amd import functional, pwl
    from gnan_amd.functional import _fpwl_moments
    monkeypatch.setattr(functional, "MOMENTS_FIXED_POINT", fixed)
    sd = _mlp_state(F, L, H, C, True, seed=F)
    st = _stack(sd, F, L, H, C, True)
    t = pwl.build_tables(st)
    n = 20_000
    x = torch.rand(n, F, generator=torch.Generator().manual_seed(1)) * 4 - 2
    g = torch.randn(n, C if sum_features else F * C, generator=torch.Generator().manual_seed(2))
    tc = pwl.PwlTables(*[q.cpu() if torch.is_tensor(q) else q for q in t])
    want = pwl.moments_reference(x, g, tc, sum_features)
    xd, gd = x.to(DEV), g.to(DEV)
    got = _fpwl_moments(xd, t, gd, sum_features).cpu().double()
    assert_rule(got, want, None, "moments")                 # float64 reference of the same sums: the floor
    scale = float(want.abs().max())
    if fixed:                                   # integer accumulation: bit-reproducible, and exact to ~2^-40 of the largest term
        assert torch.equal(_fpwl_moments(xd, t, gd, sum_features), _fpwl_moments(xd, t, gd, sum_features))
        assert float((got - want).abs().max()) <= 1e-6 * scale
        tiny = _fpwl_moments(xd, t, gd * 1e-30, sum_features).cpu().double()          # the scale follows the gradient's size
        assert float((tiny - want * 1e-30).abs().max()) <= 2e-6 * scale * 1e-30
        assert float(_fpwl_moments(xd, t, gd * 0, sum_features).abs().max()) == 0.0
        # the C = 1 kernel (gradient read next to x, anchor tracked by the search, one-fma fixed-point conversion) and the
        # general kernel add the same integers: identical bins, also for gradient rows that cannot be read as quads
        monkeypatch.setattr(functional, "MOMENTS_GENERAL", True)        # gnan_fpwl_args.flags & GNAN_FPWL_MOMENTS_GENERAL
        general = _fpwl_moments(xd, t, gd, sum_features, raw=True)[0]
        monkeypatch.setattr(functional, "MOMENTS_GENERAL", False)
        assert torch.equal(_fpwl_moments(xd, t, gd, sum_features, raw=True)[0], general)
        if not sum_features and C == 1:
            wide = torch.zeros(n, F + 3, device=DEV)
            wide[:, 1:F + 1] = gd
            assert torch.equal(_fpwl_moments(xd, t, wide[:, 1:F + 1], sum_features, raw=True)[0], general)
        if sum_features and C == 1 and t.features_per_group % 4 == 0:
            # the forward pass can keep the piece of every look-up (a byte each); the moment kernel then skips its search
            kept = []
            functional._fpwl_launch(xd, t, True, located=kept)
            fg = t.features_per_group
            assert len(kept) == 1 and kept[0].dtype == torch.uint8 and kept[0].shape == ((F + fg - 1) // fg, n, fg)   # group-major
            off, anchor = t.off.cpu().long(), t.anchor.cpu()
            for k in (0, F - 1):
                want_piece = torch.searchsorted(anchor[off[k]:off[k + 1]][1:].contiguous(), x[:, k].contiguous(), right=True)
                assert torch.equal(kept[0][k // fg, :, k % fg].cpu().long(), want_piece)
            # with kept pieces the kernel bins sum g and sum g * x (nothing to wait for between its atomics) and subtracts
            # anchor * sum g once per piece and workgroup: the same M0 bit for bit, M1 to float64 rounding of that product
            with_kept = _fpwl_moments(xd, t, gd, True, raw=True, located=kept)[0]
            assert torch.equal(with_kept[:, 0], general[:, 0])
            m1_scale = float(general[:, 1].abs().max())
            assert float((with_kept[:, 1] - general[:, 1]).abs().max()) <= 1e-6 * max(m1_scale, 1.0)     # (float32 rounding of the products g * (x - a) the general kernel bins)


@pytest.mark.parametrize("F,L,H,C,bias,n", [(3, 3, 8, 1, True, 203), (20, 3, 64, 3, True, 1000), (7, 3, 33, 7, False, 5),
                                             (15, 3, 64, 1, False, 31), (2, 3, 64, 8, True, 1), (129, 3, 64, 1, True, 3000),
                                             (9, 2, 64, 2, True, 77), (4, 2, 20, 5, False, 300),
                                             (2, 3, 64, 2, True, 9001), (3, 2, 16, 1, True, 5000)])   # node ranges
@pytest.mark.parametrize("sum_features", [True, False])
def test_small_batch_backward_kernel_vs_autograd(F, L, H, C, bias, n, sum_features, monkeypatch):
    """gnan_fmlp_bwd (one workgroup per feature, gradients accumulated in registers) == the batched-GEMM restatement
    differentiated by torch, and both == autograd through the float64 oracle; bit-identical from run to run."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_AUTO)
    sd = _mlp_state(F, L, H, C, bias, seed=F * 3 + H + C)
    x = (torch.rand(n, F, generator=torch.Generator().manual_seed(3)) * 4 - 2).to(DEV)
    width = C if sum_features else F * C
    gup = torch.randn(n, width, generator=torch.Generator().manual_seed(4)).to(DEV)
    got = {}
    for tag, on in (("hip", True), ("torch", False), ("hip2", True)):
        monkeypatch.setattr(functional, "HIP_SMALL_BACKWARD", on)
        st = _stack(sd, F, L, H, C, bias)
        leaves = [t for t in st[:6] if t is not None]
        for t in leaves:
            t.requires_grad_(True)
        out = feature_mlps(x, st, sum_features)
        got[tag] = torch.autograd.grad(out, leaves, gup)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = O.feature_mlps(x.cpu().double(), sd64)                      # [n, F, C]
    ref = ref.sum(1) if sum_features else ref.reshape(n, -1)
    ref.backward(gup.cpu().double())
    scale = max(float(v.grad.abs().max()) for v in sd64.values())
    for a, b, c in zip(got["hip"], got["torch"], got["hip2"]):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= TWO_FLOORS * max(scale, 1e-30), (float((a - b).abs().max()), scale)      # two routes
        assert torch.equal(a, c)
    # against the oracle's autograd: first-layer weights of all features
    w1 = torch.stack([sd64[f"fs.{k}.0.weight"].grad[:, 0] for k in range(F)])
    assert O.rel_err(got["hip"][0].cpu(), w1) <= 1e-5


@pytest.mark.parametrize("F,L,H,C,bias,n", [(5, 3, 64, 1, True, 4000), (3, 3, 33, 40, True, 900), (7, 3, 8, 3, False, 2500),
                                             (4, 2, 100, 5, True, 3000), (2, 2, 16, 64, False, 700), (1, 3, 64, 2, True, 50),
                                             (129, 3, 64, 1, True, 2000),
                                             (3, 3, 16, 100, True, 800), (2, 2, 32, 130, False, 600)])   # > 64 channels: chunks
@pytest.mark.parametrize("sum_features", [True, False])
def test_table_path_parameter_gradients_kernel(F, L, H, C, bias, n, sum_features, monkeypatch):
    """gnan_fpwl_param_grads (analytic reverse passes per piece, float64, one workgroup per feature) == the torch route
    (two probe points per piece through the batched MLP) == autograd through the float64 oracle; bit-reproducible."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    sd = _mlp_state(F, L, H, C, bias, seed=F + 5 * H + C)
    x = (torch.rand(n, F, generator=torch.Generator().manual_seed(3)) * 4 - 2).to(DEV)
    width = C if sum_features else F * C
    gup = torch.randn(n, width, generator=torch.Generator().manual_seed(4)).to(DEV)
    got = {}
    for tag, on in (("hip", True), ("torch", False), ("hip2", True)):
        monkeypatch.setattr(functional, "HIP_TABLE_GRADS", on)
        st = _stack(sd, F, L, H, C, bias)
        leaves = [t for t in st[:6] if t is not None]
        for t in leaves:
            t.requires_grad_(True)
        out = feature_mlps(x, st, sum_features)
        got[tag] = torch.autograd.grad(out, leaves, gup)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = O.feature_mlps(x.cpu().double(), sd64)
    ref = ref.sum(1) if sum_features else ref.reshape(n, -1)
    ref.backward(gup.cpu().double())
    scale = max(float(v.grad.abs().max()) for v in sd64.values())
    for a, b, c in zip(got["hip"], got["torch"], got["hip2"]):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1e-5 * scale, (float((a - b).abs().max()), scale)
        assert torch.equal(a, c)
    last = 3 * (L - 1)
    names = [("0.weight", lambda t: t[:, 0]), ("0.bias", None)] if bias else [("0.weight", lambda t: t[:, 0])]
    want = {"w_first": torch.stack([sd64[f"fs.{k}.0.weight"].grad[:, 0] for k in range(F)]),
            "w_last": torch.stack([sd64[f"fs.{k}.{last}.weight"].grad for k in range(F)])}
    leaves_names = [nm for nm, t in zip(("w_first", "b_first", "w_mid", "b_mid", "w_last", "b_last"), _stack(sd, F, L, H, C, bias)[:6])
                    if t is not None]
    by_name = dict(zip(leaves_names, got["hip"]))
    assert float((by_name["w_first"].cpu().double() - want["w_first"]).abs().max()) <= 1e-5 * scale
    assert float((by_name["w_last"].cpu().double() - want["w_last"]).abs().max()) <= 1e-5 * scale
    if L == 3:
        w_mid = torch.stack([sd64[f"fs.{k}.3.weight"].grad for k in range(F)])
        assert float((by_name["w_mid"][0].cpu().double() - w_mid).abs().max()) <= 1e-5 * scale
        if bias:
            b_mid = torch.stack([sd64[f"fs.{k}.3.bias"].grad for k in range(F)])
            assert float((by_name["b_mid"][0].cpu().double() - b_mid).abs().max()) <= 1e-5 * scale
    if bias:
        b_first = torch.stack([sd64[f"fs.{k}.0.bias"].grad for k in range(F)])
        b_last = torch.stack([sd64[f"fs.{k}.{last}.bias"].grad for k in range(F)])
        assert float((by_name["b_first"].cpu().double() - b_first).abs().max()) <= 1e-5 * scale
        assert float((by_name["b_last"].cpu().double() - b_last).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("fixed", [True, False])
def test_table_path_parameter_gradients_edge_cases(fixed, monkeypatch):
    """gnan_fpwl_param_grads on the float moments as well as the fixed-point ones; an all-zero upstream gradient (every piece
    empty: nothing is walked) gives exact zeros; a single node; nodes far outside the kinks (only the two outer rays live)."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    monkeypatch.setattr(functional, "MOMENTS_FIXED_POINT", fixed)
    F, L, H, C = 6, 3, 16, 2
    sd = _mlp_state(F, L, H, C, True, seed=33)
    for n, spread, gscale in [(700, 4.0, 1.0), (700, 4.0, 0.0), (1, 4.0, 1.0), (300, 1e4, 1.0)]:
        x = ((torch.rand(n, F, generator=torch.Generator().manual_seed(n)) - 0.5) * spread).to(DEV)
        gup = (torch.randn(n, F * C, generator=torch.Generator().manual_seed(4)) * gscale).to(DEV)
        got = {}
        for tag, on in (("hip", True), ("torch", False)):
            monkeypatch.setattr(functional, "HIP_TABLE_GRADS", on)
            st = _stack(sd, F, L, H, C, True)
            leaves = [t for t in st[:6] if t is not None]
            for t in leaves:
                t.requires_grad_(True)
            got[tag] = torch.autograd.grad(feature_mlps(x, st, False), leaves, gup)
        scale = max(float(t.abs().max()) for t in got["torch"])
        for a, b in zip(got["hip"], got["torch"]):
            if gscale == 0.0:
                assert float(a.abs().max()) == 0.0
            else:
                assert float((a - b).abs().max()) <= TWO_FLOORS * scale, (n, spread, float((a - b).abs().max()), scale)   # two routes


@pytest.mark.parametrize("F,L,H,C,sum_features", [(129, 3, 32, 40, True), (20, 3, 16, 7, True), (9, 2, 24, 2, True),
                                                   (33, 3, 16, 64, True), (12, 3, 16, 5, False), (70, 3, 8, 17, False),
                                                   (20, 3, 16, 100, True), (6, 3, 8, 130, False),    # > 64 channels: chunks
                                                   (32, 3, 16, 13, True),
                                                   (20, 3, 16, 34, True), (12, 3, 16, 41, False), (16, 3, 16, 42, True)])   # 33..42: a pair of channels per lane
def test_two_phase_lookup_for_several_channels(F, L, H, C, sum_features, monkeypatch):
    """csrc/fpwl_rows.hip: piece / dx located once per (node, feature) (bit-exact index work against the reference search),
    forward rows == the thread-per-node kernel (same arithmetic per term; the feature sum in the same order when a thread
    owns one feature) == float64 oracle; moments == the general fixed-point kernel, the same integers."""
    from gnan_amd import _lib, functional, pwl
    from gnan_amd.functional import _fpwl_launch, _fpwl_moments
    monkeypatch.setattr(functional, "FPWL_ROWS_MIN_NODES", 1)
    monkeypatch.setattr(functional, "FPWL_ROWS_MIN_CHANNELS", 2)
    monkeypatch.setattr(functional, "FPWL_ROWS_MIN_CHANNELS_LARGE", 2)
    monkeypatch.setattr(functional, "SUM_VIA_FEATURES_MAX_NODES", 0)     # (the small-graph detour around the summing kernel)
    sd = _mlp_state(F, L, H, C, True, seed=F + C)
    st = _stack(sd, F, L, H, C, True)
    t = pwl.build_tables(st)
    n = 5000
    x = (torch.rand(n, F, generator=torch.Generator().manual_seed(1)) * 4 - 2).to(DEV)
    x[:, 0] = 0.25                                                       # a constant column: every node in the same piece
    g = torch.randn(n, C if sum_features else F * C, generator=torch.Generator().manual_seed(2)).to(DEV)
    # phase 1 against the definition
    a = functional._fpwl_args(x, t, sum_features)
    piece, dx = functional._fpwl_locate(x, t, a)
    if F % 16 in (0, 1, 4):                                               # the tree-search kernel of large batches == the sorted-array one
        xb = x.repeat(53, 1)[:262144]
        pb, db = functional._fpwl_locate(xb, t, functional._fpwl_args(xb, t, sum_features))
        assert torch.equal(pb[:n], piece) and torch.equal(db[:n], dx) and torch.equal(pb[n:2 * n], piece)
    off, anchor = t.off.cpu().long(), t.anchor.cpu()
    xc = x.cpu()
    for k in (0, 1, F // 2, F - 1):
        seg = anchor[off[k]:off[k + 1]]
        want = off[k] + torch.searchsorted(seg[1:].contiguous(), xc[:, k].contiguous(), right=True)
        assert torch.equal(piece[:, k].cpu().long(), want)
        assert torch.equal(dx[:, k].cpu(), xc[:, k] - anchor[want])
    out = {}
    for rows in (True, False):
        monkeypatch.setattr(functional, "FPWL_ROWS", rows)
        out[rows] = (_fpwl_launch(x, t, sum_features), _fpwl_moments(x, t, g, sum_features, raw=True)[0])
    assert torch.equal(out[True][1], out[False][1])                      # the same integers in the bins
    truth = O.feature_mlps(x.cpu().double(), {k: v.double() for k, v in sd.items()})          # [n, F, C]
    truth = truth.sum(1) if sum_features else truth.reshape(n, F * C)
    assert O.rel_err(out[True][0].cpu(), truth) <= 1e-5
    if t.features_per_group == 1 or not sum_features:
        assert torch.equal(out[True][0], out[False][0])
    else:
        assert O.rel_err(out[True][0].cpu(), out[False][0].cpu().double()) <= 2e-6


@pytest.mark.parametrize("sum_features", [True, False])
def test_tables_too_large_for_lds_use_the_two_phase_kernels(sum_features):
    """172 output channels x ~130 pieces: not even one feature's tables fit the LDS image of the thread-per-node kernels
    (before round 2 such models fell back to the lane kernel).  The planner still returns tables, the two-phase kernels
    evaluate them from global memory at any batch size: look-up == float64 oracle, moments == the torch restatement,
    and the whole autograd path == oracle autograd."""
    from gnan_amd import functional, pwl
    from gnan_amd.functional import _fpwl_launch, _fpwl_moments, feature_mlps
    F, L, H, C, n = 3, 3, 64, 172, 3000
    sd = _mlp_state(F, L, H, C, True, seed=9)
    st = _stack(sd, F, L, H, C, True)
    t = pwl.build_tables(st)
    assert t is not None and pwl.oversize(t) and t.features_per_group == 1
    gen = torch.Generator().manual_seed(4)
    x = torch.rand(n, F, generator=gen) * 4 - 2
    g = torch.randn(n, C if sum_features else F * C, generator=gen)
    out = _fpwl_launch(x.to(DEV), t, sum_features)
    truth = O.feature_mlps(x.double(), {k: v.double() for k, v in sd.items()})
    truth = truth.sum(1) if sum_features else truth.reshape(n, F * C)
    assert O.rel_err(out.cpu(), truth) <= 1e-5
    tc = pwl.PwlTables(*[q.cpu() if torch.is_tensor(q) else q for q in t])
    want = pwl.moments_reference(x, g, tc, sum_features)
    got = _fpwl_moments(x.to(DEV), t, g.to(DEV), sum_features).cpu().double()
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    # end to end through the autograd function (whatever strategy AUTO picks for this batch size)
    params = [None if q is None else q.clone().requires_grad_(True) for q in st[:6]]
    y = feature_mlps(x.to(DEV), functional.StackedMLP(*params, *st[6:]), sum_features)
    assert O.rel_err(y.detach().cpu(), truth) <= 1e-5


@pytest.mark.parametrize("seed", range(10))
def test_table_path_agrees_with_oracle_on_random_shapes(seed, monkeypatch):
    """Seeded random models (features, depth, width, channels, batch size, gradient layout) through the table path's
    dispatcher — fast / ragged / general / two-phase kernels, kept pieces, chunked parameter gradients — against float64
    oracle autograd: outputs and parameter gradients by the rule (1e-5 of the largest, or the float32 oracle's own error).
    (Fixed seeds: a wider sweep — 60 seeds — fails once, seed 19, identically on EVERY kernel route including round 1's: a
    node whose x lies within an ulp of a kink of its shape function falls on the other side of the float32-rounded
    breakpoint, and the ReLU subgradient there is a choice — 6e-4 of the largest gradient, the same ambiguity the float32
    reference has against float64.)"""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    rng = np.random.default_rng(3000 + seed)
    F = int(rng.choice([1, 3, 8, 16, 17, 33, 64, 70, 129]))
    L = int(rng.choice([2, 3]))
    H = int(rng.choice([4, 16, 33, 64]))
    C = int(rng.choice([1, 1, 1, 2, 5, 9, 40, 70]))
    n = int(rng.choice([257, 3000, 9000]))
    bias, sum_features = bool(rng.random() < 0.7), bool(rng.random() < 0.6)
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    if rng.random() < 0.5:                                               # let the two-phase kernels take small batches too
        monkeypatch.setattr(functional, "FPWL_ROWS_MIN_NODES", 1)
        monkeypatch.setattr(functional, "SUM_VIA_FEATURES_MAX_NODES", 0)
    sd = _mlp_state(F, L, H, C, bias, seed=seed)
    st = _stack(sd, F, L, H, C, bias)
    leaves = [t for t in st[:6] if t is not None]
    for t in leaves:
        t.requires_grad_(True)
    gen = torch.Generator().manual_seed(seed)
    x = torch.rand(n, F, generator=gen) * 4 - 2
    if F > 1:
        x[:, -1] = 1.0                                                   # the reference's ones column
    gup = torch.randn(n, C if sum_features else F * C, generator=gen)
    out = feature_mlps(x.to(DEV), st, sum_features)
    got = torch.autograd.grad(out, leaves, gup.to(DEV))
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = O.feature_mlps(x.double(), sd64)
    ref = ref.sum(1) if sum_features else ref.reshape(n, -1)
    assert O.rel_err(out.detach().cpu(), ref.detach()) <= 1e-5
    ref.backward(gup.double())
    last = 3 * (L - 1)
    want = {"w_first": torch.stack([sd64[f"fs.{k}.0.weight"].grad[:, 0] for k in range(F)]),
            "w_last": torch.stack([sd64[f"fs.{k}.{last}.weight"].grad for k in range(F)])}
    if bias:
        want["b_first"] = torch.stack([sd64[f"fs.{k}.0.bias"].grad for k in range(F)])
        want["b_last"] = torch.stack([sd64[f"fs.{k}.{last}.bias"].grad for k in range(F)])
    if L == 3:
        want["w_mid"] = torch.stack([sd64[f"fs.{k}.3.weight"].grad for k in range(F)]).unsqueeze(0)
        if bias:
            want["b_mid"] = torch.stack([sd64[f"fs.{k}.3.bias"].grad for k in range(F)]).unsqueeze(0)
    names = [nm for nm, t in zip(("w_first", "b_first", "w_mid", "b_mid", "w_last", "b_last"), st[:6]) if t is not None]
    def oracle32():
        sd32 = {k: v.float().requires_grad_(True) for k, v in sd.items()}
        r = O.feature_mlps(x, sd32)
        (r.sum(1) if sum_features else r.reshape(n, -1)).backward(gup)
        w = {"w_first": torch.stack([sd32[f"fs.{k}.0.weight"].grad[:, 0] for k in range(F)]),
             "w_last": torch.stack([sd32[f"fs.{k}.{last}.weight"].grad for k in range(F)])}
        if bias:
            w["b_first"] = torch.stack([sd32[f"fs.{k}.0.bias"].grad for k in range(F)])
            w["b_last"] = torch.stack([sd32[f"fs.{k}.{last}.bias"].grad for k in range(F)])
        if L == 3:
            w["w_mid"] = torch.stack([sd32[f"fs.{k}.3.weight"].grad for k in range(F)]).unsqueeze(0)
            if bias:
                w["b_mid"] = torch.stack([sd32[f"fs.{k}.3.bias"].grad for k in range(F)]).unsqueeze(0)
        return w
    assert_grads_rule(dict(zip(names, got)), want, oracle32, (F, L, H, C, n, sum_features))


@pytest.mark.parametrize("n,width,gscale", [(1000, 3, 1.0), (70000, 1, 1e-12), (5, 64, 1e20), (100, 2, 0.0)])
def test_moment_scales_kernel(n, width, gscale):
    """gnan_fpwl_moment_scales == the framework formula it replaced (powers of two from max|grad| and max|x - anchor|)."""
    from gnan_amd import _lib
    g = (torch.randn(n, width + 2, generator=torch.Generator().manual_seed(n)) * gscale).to(DEV)[:, :width]     # strided rows
    anchor = (torch.randn(300, generator=torch.Generator().manual_seed(1)) * 5).to(DEV)
    xmax = torch.tensor(3.25, dtype=torch.float64, device=DEV)
    bits = 61 - max(1, (max(n, 2) - 1).bit_length())
    ws = _lib.MOMENT_SCALES_WORKSPACE_BYTES
    out = torch.empty(2 + ws // 8, dtype=torch.float64, device=DEV)
    cleared = torch.full((1000,), 7, dtype=torch.int64, device=DEV)              # the moment accumulators: zeroed by the same pass
    padded = torch.cat([anchor, torch.full((50,), float("inf"), device=DEV)])     # a buffer of full capacity: the tail is not data
    n_real = torch.tensor([anchor.numel()], dtype=torch.int32, device=DEV)
    a = _lib.MomentScalesArgs(grad=_lib.ptr(g), n=n, width=width, bits=bits, grad_stride=g.stride(0), anchor=_lib.ptr(padded),
                              T=padded.numel(), n_anchors=_lib.ptr(n_real), x_abs_max=_lib.ptr(xmax), workspace=_lib.ptr(out[2:]),
                              workspace_bytes=ws, scales=_lib.ptr(out), zero=_lib.ptr(cleared), zero_bytes=999 * 8)
    _lib.check(_lib.lib().gnan_fpwl_moment_scales(a, _lib.stream_of(g)), "gnan_fpwl_moment_scales")
    assert int(cleared[:999].abs().sum()) == 0 and int(cleared[999]) == 7
    tiny = torch.finfo(torch.float64).tiny
    g_max = g.abs().max().double().clamp_min(tiny)
    d_max = (xmax + anchor.abs().max().double()).clamp_min(tiny)
    eb = min(bits, 50)                          # the library caps the exponent: every term stays below 2^51 (fixed_bits)
    e = torch.stack([torch.floor(eb - torch.log2(g_max)), torch.floor(eb - torch.log2(g_max * d_max))])
    want = torch.exp2(e.clamp(-1000.0, 1000.0))
    assert torch.equal(out[:2].cpu(), want.cpu()), (out[:2], want)
    gn = g.clone()
    gn[n // 2, 0] = float("nan")
    a = _lib.MomentScalesArgs(grad=_lib.ptr(gn), n=n, width=width, bits=bits, grad_stride=gn.stride(0), anchor=_lib.ptr(anchor),
                              T=anchor.numel(), n_anchors=None, x_abs_max=_lib.ptr(xmax), workspace=_lib.ptr(out[2:]),
                              workspace_bytes=ws, scales=_lib.ptr(out))
    _lib.check(_lib.lib().gnan_fpwl_moment_scales(a, _lib.stream_of(g)), "gnan_fpwl_moment_scales")
    assert bool(torch.isnan(out[:2]).all())
    a.workspace_bytes = 8                                                        # too small a workspace is refused, not overrun
    assert _lib.lib().gnan_fpwl_moment_scales(a, _lib.stream_of(g)) != 0
    a.workspace_bytes = ws
    _lib.check(_lib.lib().gnan_fpwl_moment_scales(a, _lib.stream_of(g)), "gnan_fpwl_moment_scales")
    assert bool(torch.isnan(out[:2]).all())


@pytest.mark.parametrize("algo", ["auto", "pwl"])
def test_expanded_upstream_gradient(algo, monkeypatch):
    """``feature_mlps(...).sum(0)`` sends back an EXPANDED gradient (strides (0, 1)); both backward routes — the
    per-feature kernel of small batches and the moment kernel of the table path — must take it (they read rows)."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_AUTO if algo == "auto" else _lib.FMLP_PWL)
    F, L, H, C, n = 6, 3, 16, 2, 500
    sd = _mlp_state(F, L, H, C, True, seed=11)
    x = (torch.rand(n, F, generator=torch.Generator().manual_seed(3)) * 4 - 2).to(DEV)
    wcol = torch.randn(F * C, generator=torch.Generator().manual_seed(5)).to(DEV)
    st = _stack(sd, F, L, H, C, True)
    leaves = [t for t in st[:6] if t is not None]
    for t in leaves:
        t.requires_grad_(True)
    out = feature_mlps(x, st, False)                                  # [n, F*C]
    got = torch.autograd.grad((out.sum(0) * wcol).sum(), leaves)      # d/d out = wcol expanded over the rows
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = O.feature_mlps(x.cpu().double(), sd64).reshape(n, -1)
    (ref.sum(0) * wcol.cpu().double()).sum().backward()
    w1 = torch.stack([sd64[f"fs.{k}.0.weight"].grad[:, 0] for k in range(F)])
    wl = torch.stack([sd64[f"fs.{k}.6.weight"].grad for k in range(F)])
    assert O.rel_err(got[0].cpu(), w1) <= 1e-5
    assert O.rel_err(got[4].cpu(), wl) <= 1e-5


@pytest.mark.parametrize("F,L,bias", [(21, 3, True), (33, 2, False), (5, 3, True)])
def test_ragged_feature_counts_are_padded(F, L, bias, monkeypatch):
    """F = raw features + the ones column is rarely a multiple of 16: large inputs are evaluated with all-zero shape
    functions appended (whole 16-feature groups, aligned rows).  Same values and gradients as the unpadded evaluation,
    in both result layouts, with the fused column sums, and bf16 rows become available."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    H, C, n = 16, 1, 40000
    sd = _mlp_state(F, L, H, C, bias, seed=F + L)
    x = torch.rand(n, F, generator=torch.Generator().manual_seed(9)).to(DEV)
    gsum = torch.randn(n, C, generator=torch.Generator().manual_seed(1)).to(DEV)
    gper = torch.randn(n, F, generator=torch.Generator().manual_seed(2)).to(DEV)
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    results = {}
    for tag, pad in (("plain", 0), ("padded", 16)):
        monkeypatch.setattr(functional, "PAD_FEATURES", pad)
        monkeypatch.setattr(functional, "PAD_MIN_WORK", 1)
        st = _stack(sd, F, L, H, C, bias)
        leaves = [t for t in st[:6] if t is not None]
        for t in leaves:
            t.requires_grad_(True)
        s_out = feature_mlps(x, st, True)
        g_sum = torch.autograd.grad(s_out, leaves, gsum)
        p_out, tot = feature_mlps(x, st, False, return_total=True)
        assert p_out.shape == (n, F) and tot.shape == (F,)
        g_per = torch.autograd.grad(p_out, leaves, gper)
        wide = feature_mlps(x, st, False, pad_ok=True)
        results[tag] = (s_out.detach(), p_out.detach(), tot, g_sum, g_per, wide.detach())
    a, b = results["plain"], results["padded"]
    assert b[5].shape == (n, (F + 15) // 16 * 16) and float(b[5][:, F:].abs().max()) == 0.0 and torch.equal(b[5][:, :F], b[1])
    assert a[5].shape == (n, F)
    assert O.rel_err(b[0].cpu(), a[0].double().cpu()) <= 2e-6 and O.rel_err(b[1].cpu(), a[1].double().cpu()) <= 2e-6
    assert O.rel_err(b[2].cpu(), a[2].double().cpu()) <= 2e-6
    for ga, gb in zip(a[3] + a[4], b[3] + b[4]):
        assert ga.shape == gb.shape and O.rel_err(gb.cpu(), ga.double().cpu()) <= 1e-5
    monkeypatch.setattr(functional, "PAD_FEATURES", 16)
    with torch.no_grad():
        rows16 = feature_mlps(x, _stack(sd, F, L, H, C, bias), False, out_dtype=torch.bfloat16, pad_ok=True)
    assert rows16.dtype == torch.bfloat16 and torch.equal(rows16[:, :F], a[1].to(torch.bfloat16))


@pytest.mark.parametrize("C", [1, 3])
def test_speculative_lookup_survives_a_wrong_guess(C, monkeypatch):
    """The look-up is queued with the LAST forward's table sizes before this forward's are known: same output when the
    guess was right, when there was none, and when it was far too small (the look-up is then queued again)."""
    from gnan_amd import _lib, functional, pwl
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    F, L, H, n = 32, 3, 32, 70000
    sd = _mlp_state(F, L, H, C, True, seed=77)
    st = _stack(sd, F, L, H, C, True)
    x = torch.rand(n, F, generator=torch.Generator().manual_seed(5)).to(DEV)
    monkeypatch.setattr(functional, "SPECULATIVE_LOOKUP", False)
    with torch.no_grad():
        want, want_tot = feature_mlps(x, st, False, return_total=True)
    monkeypatch.setattr(functional, "SPECULATIVE_LOOKUP", True)
    pwl._LAST_PLAN.clear()
    with torch.no_grad():
        first, first_tot = feature_mlps(x, st, False, return_total=True)          # no guess yet
        assert len(pwl._LAST_PLAN) == 1
        second, second_tot = feature_mlps(x, st, False, return_total=True)        # right guess
        key = next(iter(pwl._LAST_PLAN))
        pwl._LAST_PLAN[key] = (5, 16, 40)                                         # far too small: depth 3 bits, 40 pieces per group
        third, third_tot = feature_mlps(x, st, False, return_total=True)
        pwl._LAST_PLAN[key] = (5, 4, 4000)                                        # another grouping, too shallow a search
        fourth = feature_mlps(x, st, True)
    for got, tot in ((first, first_tot), (second, second_tot), (third, third_tot)):
        assert torch.equal(got, want) and torch.equal(tot, want_tot)
    assert O.rel_err(fourth.cpu(), want.double().view(n, F, C).sum(1).cpu()) <= 1e-6


def test_prefetched_tables_give_the_same_forward(monkeypatch):
    """functional.TablePrefetch: the tables built ahead of time on a side stream are the tables the in-line build makes;
    outputs, fused column sums and bf16 rows are bit-identical, several builds may be in flight, and gradients refuse them."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import TablePrefetch, feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    F, L, H, C, n = 32, 3, 64, 1, 50_000
    sd = _mlp_state(F, L, H, C, True, seed=21)
    st = _stack(sd, F, L, H, C, True)
    x = torch.rand(n, F, generator=torch.Generator().manual_seed(3)).to(DEV)
    pre = TablePrefetch(st)
    assert pre.applies
    with torch.no_grad():
        want, want_total = feature_mlps(x, st, False, return_total=True, total_rows=n // 2)
        first, second = pre.launch(), pre.launch()                     # two builds in flight
        got, got_total = feature_mlps(x, st, False, return_total=True, total_rows=n // 2, tables=first)
        again = feature_mlps(x, st, True, tables=second)
        assert torch.equal(got, want) and torch.equal(got_total, want_total)
        assert torch.equal(again, feature_mlps(x, st, True))
        b16 = feature_mlps(x, st, False, out_dtype=torch.bfloat16, tables=pre.launch())
        assert torch.equal(b16, feature_mlps(x, st, False, out_dtype=torch.bfloat16))
    for t in st[:6]:
        if t is not None:
            t.requires_grad_(True)
    with pytest.raises(_lib.GnanHipError, match="inference"):
        feature_mlps(x, st, False, tables=pre.launch())


@pytest.mark.parametrize("n,F", [(70_000, 64), (300_000, 16), (5000, 32)])
def test_fused_column_sums_of_table_lookup(n, F, monkeypatch):
    """feature_mlps(return_total=True): the look-up kernel's fused column sums == a separate pass over its output."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import column_sums, feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    sd = _mlp_state(F, 3, 16, 1, True, seed=F)
    st = _stack(sd, F, 3, 16, 1, True)
    x = torch.rand(n, F, device=DEV) * 2 - 1
    out, total = feature_mlps(x, st, False, return_total=True)
    want = out.double().sum(0)
    assert float((total.double() - want).abs().max()) <= 1e-6 * float(out.abs().sum(0).max())
    assert torch.equal(total, feature_mlps(x, st, False, return_total=True)[1])       # fixed reduction order
    for rows in (0, 1, 777, n // 2 + 3, n):                                          # a rank's owned rows ahead of its halo
        out_r, total_r = feature_mlps(x, st, False, return_total=True, total_rows=rows)
        assert torch.equal(out_r, out)
        want_r = out[:rows].double().sum(0)
        assert float((total_r.double() - want_r).abs().max()) <= 1e-6 * max(1e-30, float(out.abs().sum(0).max()))
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_MFMA)                     # other strategies: separate pass
    out2, total2 = feature_mlps(x, st, False, return_total=True)
    assert torch.equal(total2, column_sums(out2))
    assert torch.equal(feature_mlps(x, st, False, return_total=True, total_rows=777)[1], column_sums(out2[:777]))


@pytest.mark.parametrize("F,L,H,C,bias", [(5, 3, 8, 1, True), (9, 3, 32, 5, False), (64, 3, 64, 1, True), (6, 3, 33, 2, True),
                                           (4, 2, 8, 3, True), (3, 3, 20, 40, True), (2, 3, 128, 2, True)])
def test_table_build_kernel_matches_torch_builder(F, L, H, C, bias, monkeypatch):
    """gnan_pwl_build (one workgroup per feature) vs the torch restatement of the same procedure, and vs the oracle."""
    from gnan_amd import pwl
    sd = _mlp_state(F, L, H, C, bias, seed=F * 7 + H)
    st = _stack(sd, F, L, H, C, bias)
    monkeypatch.setattr(pwl, "BUILD_BACKEND", "torch")
    t_ref = pwl.build_tables(st)
    monkeypatch.setattr(pwl, "BUILD_BACKEND", "auto")
    t_hip = pwl.build_tables(st)
    assert torch.equal(t_ref.off, t_hip.off), "same number of pieces per feature"
    assert float((t_ref.anchor - t_hip.anchor).abs().max()) <= 1e-6 * max(1.0, float(t_ref.anchor.abs().max()))
    x = (torch.rand(5000, F, generator=torch.Generator().manual_seed(2)) * 6 - 3)
    x[:4] = torch.tensor([0.0, 1.0, -50.0, 50.0]).unsqueeze(1)
    truth = O.feature_mlps(x.double(), {k: v.double() for k, v in sd.items()}).reshape(5000, -1)
    tc = pwl.PwlTables(*[q.cpu() if torch.is_tensor(q) else q for q in t_hip])
    assert O.rel_err(pwl.evaluate_reference(x, tc, False), truth) <= 1e-5


def test_table_build_kernel_degenerate_weights():
    """Zero biases (all first-layer kinks coincide at 0), dead units (w1 = 0) and an all-zero feature."""
    from gnan_amd import pwl
    F, L, H, C = 4, 3, 16, 2
    sd = _mlp_state(F, L, H, C, True, seed=3)
    for k in range(F):
        sd[f"fs.{k}.0.bias"].zero_()
    sd["fs.1.0.weight"][::2] = 0.0
    sd["fs.2.0.weight"].zero_()
    sd["fs.2.0.bias"].fill_(0.25)
    st = _stack(sd, F, L, H, C, True)
    t = pwl.build_tables(st)
    assert torch.isfinite(t.val).all() and torch.isfinite(t.slope).all() and torch.isfinite(t.anchor).all()
    x = torch.rand(3000, F) * 4 - 2
    truth = O.feature_mlps(x.double(), {k: v.double() for k, v in sd.items()}).reshape(3000, -1)
    tc = pwl.PwlTables(*[q.cpu() if torch.is_tensor(q) else q for q in t])
    assert O.rel_err(pwl.evaluate_reference(x, tc, False), truth) <= 1e-5


@pytest.mark.parametrize("W,dyc,use_cnt,with_rest", [(64, 64, True, True), (64, 1, True, True), (1, 1, True, True),
                                                      (8, 2, False, True), (6, 6, True, False), (16, 16, False, False),
                                                      (300, 4, True, True)])
def test_fused_table_gradient_equals_shell_sums_route(W, dyc, use_cnt, with_rest):
    """gnan_spmm_lut_grad (one pass, no [n, D, W] tensor) == shell sums contracted with dY in float64, for rows, hub
    rows, row subsets, the broadcast gradient of the fused read-out (dy_channels < W) and both reduction modes."""
    from gnan_amd.aggregate import lut_grad_launch, shell_sums_launch
    rng = np.random.default_rng(W * 3 + dyc)
    n, K = 3000, 2
    D = K + 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(7, 700), (100, 5000), (2999, 2100)])
    g = _graph(rowptr, col, code, n, D)
    S = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    dummy = torch.zeros((D, 1), device=DEV)
    for ids in (None, torch.from_numpy(rng.integers(0, n, 500).astype(np.int32)).to(DEV)):
        n_out = n if ids is None else 500
        dY = torch.from_numpy(rng.standard_normal((n_out, dyc)).astype(np.float32)).to(DEV)
        T = shell_sums_launch(g, S, dummy, with_rest, ids).double()                       # [n_out, D, W]
        want = (T * dY.double().repeat(1, W // dyc).unsqueeze(1)).sum(2)                  # [n_out, D]
        if use_cnt:
            cnt = g.cnt if ids is None else g.cnt[ids.long()]
            want = want / cnt.clamp_min(1).double()
        rows = lut_grad_launch(g, S, dY, D, use_cnt, with_rest, ids, None, False)
        assert rows.shape == (n_out, D, 1)
        assert_rule(rows[..., 0], want, None, "per-row table gradient")          # float64 contraction of the same shell sums: the floor
        total = lut_grad_launch(g, S, dY, D, use_cnt, with_rest, ids, None, True)
        assert total.shape == (D, 1)
        assert float((total[:, 0].double() - want.sum(0)).abs().max()) <= 1e-5 * float(want.abs().sum(0).max())   # (against the sum of magnitudes: the terms cancel)
        assert torch.equal(total, lut_grad_launch(g, S, dY, D, use_cnt, with_rest, ids, None, True))   # fixed order


@pytest.mark.parametrize("n,D,W,dyc,use_cnt", [(30, 7, 1, 1, True), (90, 12, 7, 7, True), (700, 40, 5, 5, False), (1500, 200, 3, 3, True),
                                                (333, 255, 8, 2, True), (2708, 15, 7, 7, True)])
def test_dense_table_gradient_in_one_pass(n, D, W, dyc, use_cnt, monkeypatch):
    """gnan_spmm_lut_grad on the DENSE layout (dense_lut_grad_kernel: one wave per row, dot products binned by hop code in LDS,
    up to 256 shells) == shell sums contracted with dY in float64; row subsets; the broadcast gradient of the fused read-out;
    bit-reproducible — and rho_aggregate's backward through it == the shell-sum route == float64 oracle autograd, with the
    operand gradient of small dense graphs read from (lut, cnt) per pair instead of a per-node weight table."""
    from gnan_amd import HopGraph, functional
    from gnan_amd.aggregate import lut_grad_launch, rho_aggregate, shell_sums_launch
    rng = np.random.default_rng(n + D)
    hops = rng.integers(-1, D - 1, (n, n)).astype(np.int32)          # -1 = unreachable: the last code
    hops[np.arange(n), np.arange(n)] = 0
    nd = torch.zeros(n, n)
    nd[torch.from_numpy(hops >= 0)] = 1.0 / (torch.from_numpy(hops[hops >= 0]).float() + 1.0)
    g = HopGraph.from_dense(nd.to(DEV))
    Dg = g.n_codes
    S = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    dummy = torch.zeros((Dg, 1), device=DEV)
    for ids in (None, torch.from_numpy(rng.integers(0, n, 17).astype(np.int32)).to(DEV)):
        n_out = n if ids is None else 17
        dY = torch.from_numpy(rng.standard_normal((n_out, dyc)).astype(np.float32)).to(DEV)
        T = shell_sums_launch(g, S, dummy, False, ids).double()
        want = (T * dY.double().repeat(1, W // dyc).unsqueeze(1)).sum(2)
        if use_cnt:
            cnt = g.cnt if ids is None else g.cnt[ids.long()]
            want = want / cnt.clamp_min(1).double()
        total = lut_grad_launch(g, S, dY, Dg, use_cnt, False, ids, None, True)
        assert total.shape == (Dg, 1)
        assert float((total[:, 0].double() - want.sum(0)).abs().max()) <= 1e-5 * float(want.abs().sum(0).max())
        assert torch.equal(total, lut_grad_launch(g, S, dY, Dg, use_cnt, False, ids, None, True))
    if dyc != W:
        return
    lut0 = torch.from_numpy(rng.standard_normal((Dg, 1)).astype(np.float32)).to(DEV)
    up = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    got = {}
    for tag, fused in (("one_pass", True), ("shell_sums", False)):
        monkeypatch.setattr(aggregate, "DENSE_LUT_GRAD", fused)
        monkeypatch.setattr(aggregate, "SMALL_DENSE_ROWS", 1024 if fused else 0)
        Sx, lut = S.clone().requires_grad_(True), lut0.clone().requires_grad_(True)
        got[tag] = torch.autograd.grad(rho_aggregate(g, Sx, lut, use_cnt, with_rest=False), [Sx, lut], up)
    S64, lut64 = S.cpu().double().requires_grad_(True), lut0.cpu().double().requires_grad_(True)
    wt = lut64.unsqueeze(0).expand(n, -1, -1)
    if use_cnt:
        wt = wt / g.cnt.cpu().clamp_min(1).double().unsqueeze(-1)
    codes = torch.from_numpy(np.where(hops >= 0, hops, Dg - 1)).long()
    A = wt[torch.arange(n).unsqueeze(1), codes, 0]                         # [n, n] pair weights
    ref = torch.autograd.grad(A @ S64, [S64, lut64], up.cpu().double())
    def ref32():
        S32, l32 = S.cpu().requires_grad_(True), lut0.cpu().requires_grad_(True)
        w32 = l32.unsqueeze(0).expand(n, -1, -1)
        if use_cnt:
            w32 = w32 / g.cnt.cpu().clamp_min(1).float().unsqueeze(-1)
        return torch.autograd.grad(w32[torch.arange(n).unsqueeze(1), codes, 0] @ S32, [S32, l32], up.cpu())
    for k in range(2):                                          # each gradient against its own largest entry
        for tag in got:
            assert_rule(got[tag][k], ref[k], lambda k=k: ref32()[k], (tag, k))


@pytest.mark.parametrize("W,K,use_cnt,with_rest", [(1, 1, True, True), (1, 2, True, True), (2, 1, False, True), (3, 2, True, True),
                                                     (4, 1, True, False), (8, 2, True, True), (16, 1, True, True), (5, 2, False, False)])
def test_fused_narrow_backward_equals_two_pass_route(W, K, use_cnt, with_rest, monkeypatch):
    """gnan_spmm_bwd_narrow (operand gradient AND table gradient from one pass over the transposed adjacency) == the
    two-pass route (pre-weighted gather + gnan_spmm_lut_grad) == autograd through the float64 oracle; hub columns (sliced
    rows of the transposed graph), empty rows, padded widths; bit-reproducible."""
    from gnan_amd import functional
    from gnan_amd.aggregate import rho_aggregate
    rng = np.random.default_rng(W * 11 + K)
    n, D = 3000, K + 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(7, 700), (100, 2500)])
    col[rng.random(col.shape[0]) < 0.25] = 11                      # node 11 is listed by ~5000 rows: a hub of the transposed graph
    col[(rng.random(col.shape[0]) < 0.03)] = 12                    # ... and node 12 by ~600 (between the two hub thresholds)
    g = _graph(rowptr, col, code, n, D)
    S0 = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    lut0 = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).to(DEV)
    up = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    got = {}
    for tag, on in (("fused", True), ("two_pass", False), ("fused2", True)):
        monkeypatch.setattr(aggregate, "NARROW_FUSED_BACKWARD", on)
        S, lut = S0.clone().requires_grad_(True), lut0.clone().requires_grad_(True)
        Y = rho_aggregate(g, S, lut, use_cnt, with_rest=with_rest)
        got[tag] = torch.autograd.grad(Y, [S, lut], up)
    S64, lut64 = S0.cpu().double().requires_grad_(True), lut0.cpu().double().requires_grad_(True)
    wt = lut64.unsqueeze(0).expand(n, -1, -1)
    if use_cnt:
        wt = wt / g.cnt.cpu().clamp_min(1).double().unsqueeze(-1)
    want = O.spmm_csr(rowptr, col, code, S64, wt, with_rest=with_rest)
    dS64, dlut64 = torch.autograd.grad(want, [S64, lut64], up.cpu().double())
    def ref32():
        S32, l32 = S0.cpu().requires_grad_(True), lut0.cpu().requires_grad_(True)
        w32 = l32.unsqueeze(0).expand(n, -1, -1)
        if use_cnt:
            w32 = w32 / g.cnt.cpu().clamp_min(1).float().unsqueeze(-1)
        return torch.autograd.grad(O.spmm_csr(rowptr, col, code, S32, w32, with_rest=with_rest), [S32, l32], up.cpu())
    for k, ref in ((0, dS64), (1, dlut64)):
        assert_rule(got["fused"][k], ref, lambda k=k: ref32()[k], ("fused", k))
        assert_rule(got["two_pass"][k], ref, lambda k=k: ref32()[k], ("two_pass", k))
        assert torch.equal(got["fused"][k], got["fused2"][k])


@pytest.mark.parametrize("W,D,use_cnt,with_rest", [(1, 3, True, True), (3, 4, True, True), (4, 3, False, True), (16, 3, True, False),
                                                     (5, 2, False, False)])
def test_packed_backward_rows_kernel(W, D, use_cnt, with_rest):
    """gnan_spmm_pack_bwd_rows == the slicing restatement (tests/cpu_kernels.py): [dY / cnt(i, d) | dY / cnt(i, rest)],
    zero padded halves, strided gradient rows, counts of 0 treated as 1; empty input."""
    from gnan_amd.aggregate import pack_bwd_rows
    import cpu_kernels
    rng = np.random.default_rng(W * 7 + D)
    n = 4097
    half = 1 << max(0, (W - 1).bit_length())
    wide = torch.from_numpy(rng.standard_normal((n, W + 3)).astype(np.float32)).to(DEV)
    dY = wide[:, 1:1 + W]                                                                   # row stride W + 3
    cnt = torch.from_numpy(rng.integers(0, 50, (n, D)).astype(np.int32)).to(DEV) if use_cnt else None
    V = pack_bwd_rows(dY, cnt, D, with_rest, half)
    want = cpu_kernels.pack_bwd_rows(dY.cpu(), None if cnt is None else cnt.cpu(), D, with_rest, half)
    assert V.shape == (D, n, 2 * half)                                   # code-major
    assert torch.equal(V.cpu(), want)                                    # correctly rounded division, zero padding
    assert pack_bwd_rows(dY[:0], None if cnt is None else cnt[:0], D, with_rest, half).shape == (D, 0, 2 * half)
    ids = torch.from_numpy(rng.integers(0, n, 37)).to(DEV)                 # second copies of 37 nodes behind the n real rows
    Vh = pack_bwd_rows(dY, cnt, D, with_rest, half, hot=ids)
    assert Vh.shape == (D, n + 37, 2 * half) and torch.equal(Vh[:, :n], V) and torch.equal(Vh[:, n:], V[:, ids])
    if W == 1 and with_rest:            # the rest halves' column sum out of the same pass (real nodes only: not the hot copies)
        Vq, q = pack_bwd_rows(dY, cnt, D, with_rest, half, hot=ids, want_q_sum=True)
        assert torch.equal(Vq, Vh)
        ref = V[0, :, half].double().sum()
        assert abs(float(q[0]) - float(ref)) <= 1e-6 * float(V[0, :, half].abs().double().sum())
        assert torch.equal(q, pack_bwd_rows(dY, cnt, D, with_rest, half, hot=ids, want_q_sum=True)[1])      # fixed order


@pytest.mark.parametrize("n,D,with_rest,n_hot", [(1 << 20, 3, True, 0), ((1 << 20) + 2, 3, True, 38), ((1 << 20) + 1, 3, True, 37),
                                                   (1 << 20, 2, False, 16), (1 << 20, 4, True, 0), ((1 << 20) + 1, 3, True, 0)])
def test_packed_backward_rows_of_large_one_channel_graphs(n, D, with_rest, n_hot):
    """From 2^20 nodes on, one-channel gradients with shell counts are packed two nodes per thread (pack_bwd_pairs_kernel: 8-byte
    count loads, 16-byte stores) when every code's block starts 16-byte aligned, the odd tail and the hot copies by the one-node
    kernel: same bits as the slicing restatement either way (odd n + n_hot: the one-node kernel alone)."""
    from gnan_amd.aggregate import pack_bwd_rows
    import cpu_kernels
    rng = np.random.default_rng(n % 1000 + D)
    dY = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).to(DEV)
    cnt = torch.from_numpy(rng.integers(0, 50, (n, D)).astype(np.int32)).to(DEV)
    ids = torch.from_numpy(rng.integers(0, n, n_hot)).to(DEV) if n_hot else None
    V = pack_bwd_rows(dY, cnt, D, with_rest, 1, hot=ids)
    want = cpu_kernels.pack_bwd_rows(dY.cpu(), cnt.cpu(), D, with_rest, 1)
    assert V.shape == (D, n + n_hot, 2)
    assert torch.equal(V[:, :n].cpu(), want)
    if n_hot:
        assert torch.equal(V[:, n:], V[:, ids])
    if with_rest:
        Vq, q = pack_bwd_rows(dY, cnt, D, with_rest, 1, hot=ids, want_q_sum=True)
        assert torch.equal(Vq, V)
        ref = want[0, :, 1].double().sum()
        assert abs(float(q[0]) - float(ref)) <= 1e-6 * float(want[0, :, 1].abs().double().sum())


@pytest.mark.parametrize("W,K,with_rest", [(1, 1, True), (2, 2, True), (4, 1, False)])
def test_fused_narrow_backward_walks_sorted_copy_with_hot_columns(W, K, with_rest, monkeypatch):
    """gnan_spmm_bwd_narrow over the degree-sorted copy of the transposed adjacency with the hot packed rows appended ==
    the natural-order call: operand gradient bit for bit, table gradient to float64 round-off (its partials are added in
    processing order); both == float64 oracle autograd."""
    from gnan_amd import functional, graph as G
    from gnan_amd.aggregate import rho_aggregate
    rng = np.random.default_rng(W * 13 + K)
    n, D = 4000, K + 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(7, 700), (100, 2500)])
    col[rng.random(col.shape[0]) < 0.25] = 11                      # hub columns = hub rows of the transposed graph
    g = _graph(rowptr, col, code, n, D)
    S0 = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    lut0 = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).to(DEV)
    up = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    monkeypatch.setattr(G, "HOT_COLUMNS", 64)
    monkeypatch.setattr(G, "HOT_COLUMNS_MIN_NNZ", 0)
    monkeypatch.setattr(aggregate, "NARROW_SORTED_MIN_NNZ", 0)
    got = {}
    for tag, min_rows in (("natural", 1 << 30), ("sorted_hot", 1)):
        monkeypatch.setattr(aggregate, "DEGREE_SORTED_COPY_MIN_ROWS", min_rows)
        S, lut = S0.clone().requires_grad_(True), lut0.clone().requires_grad_(True)
        Y = rho_aggregate(g, S, lut, True, with_rest=with_rest)
        got[tag] = torch.autograd.grad(Y, [S, lut], up)
    gt = g.transposed()
    assert gt._sorted_copy_hot is not None and gt._sorted_copy_hot.n_cols == n + 64
    if W == 1:
        # one channel: spmm_bwd_hot_kernel (packed index stream, persistent workgroups, the head of the hot packed rows in
        # LDS) — ordinary rows bit for bit, hub rows add their pairs wave by wave instead of workgroup by workgroup
        hub = ((gt.rowptr[1:] - gt.rowptr[:-1]) > 64)                    # graph.LONG_ROW_THRESHOLD_NARROW
        assert int(hub.sum()) >= 1
        assert torch.equal(got["natural"][0][~hub], got["sorted_hot"][0][~hub])
        assert O.rel_err(got["sorted_hot"][0].cpu(), got["natural"][0].cpu().double()) <= 2e-6
        monkeypatch.setattr(aggregate, "NARROW_BWD_PERSISTENT", False)     # the generic kernel on the same copy
        S, lut = S0.clone().requires_grad_(True), lut0.clone().requires_grad_(True)
        plain = torch.autograd.grad(rho_aggregate(g, S, lut, True, with_rest=with_rest), [S, lut], up)
        assert torch.equal(plain[0], got["natural"][0])
        monkeypatch.setattr(aggregate, "HOT_ROWS_IN_LDS", False)           # persistent, every packed row from memory
        monkeypatch.setattr(aggregate, "NARROW_BWD_PERSISTENT", True)
        S, lut = S0.clone().requires_grad_(True), lut0.clone().requires_grad_(True)
        cold = torch.autograd.grad(rho_aggregate(g, S, lut, True, with_rest=with_rest), [S, lut], up)
        assert torch.equal(cold[0], got["sorted_hot"][0]) and torch.equal(cold[1], got["sorted_hot"][1])
    else:
        assert torch.equal(got["natural"][0], got["sorted_hot"][0])
    scale = float(got["natural"][1].abs().max())
    assert float((got["natural"][1] - got["sorted_hot"][1]).abs().max()) <= 1e-6 * scale
    S64, lut64 = S0.cpu().double().requires_grad_(True), lut0.cpu().double().requires_grad_(True)
    wt = lut64.unsqueeze(0).expand(n, -1, -1) / g.cnt.cpu().clamp_min(1).double().unsqueeze(-1)
    want = O.spmm_csr(rowptr, col, code, S64, wt, with_rest=with_rest)
    def ref32():
        S32, l32 = S0.cpu().requires_grad_(True), lut0.cpu().requires_grad_(True)
        w32 = l32.unsqueeze(0).expand(n, -1, -1) / g.cnt.cpu().clamp_min(1).float().unsqueeze(-1)
        return torch.autograd.grad(O.spmm_csr(rowptr, col, code, S32, w32, with_rest=with_rest), [S32, l32], up.cpu())
    for k, ref in enumerate(torch.autograd.grad(want, [S64, lut64], up.cpu().double())):
        assert_rule(got["sorted_hot"][k], ref, lambda k=k: ref32()[k], k)


@pytest.mark.parametrize("W,s_by_code", [(1, False), (2, False), (3, False), (4, False), (2, True)])
def test_narrow_rows_walk_sorted_copy_with_hot_columns(W, s_by_code, monkeypatch):
    """Narrow operand rows walked through the degree-sorted copy of the CSR, with and without the compact copy of the
    most listed neighbours' rows behind the operand (HopGraph.hot_columns) == natural order, bit for bit; hub rows,
    empty rows, pre-weighted (node, hop code) rows (s_by_code)."""
    from gnan_amd import functional, graph as G
    from gnan_amd.aggregate import spmm_launch
    monkeypatch.setattr(G, "HOT_COLUMNS", 64)
    monkeypatch.setattr(G, "HOT_COLUMNS_MIN_NNZ", 0)
    monkeypatch.setattr(aggregate, "DEGREE_SORTED_COPY_MIN_ROWS", 1)
    monkeypatch.setattr(aggregate, "NARROW_SORTED_MIN_NNZ", 0)
    rng = np.random.default_rng(W + 40)
    n, K = 5000, 2
    D = K + 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(7, 700), (100, 2500), (4999, 90)])
    hot = rng.random(col.shape[0]) < 0.4
    col[hot] = rng.integers(0, 50, int(hot.sum())) * 97                    # 50 nodes are listed by 40 % of the pairs
    g = _graph(rowptr, col, code, n, D)
    S = torch.from_numpy(rng.standard_normal((n * (D if s_by_code else 1), W)).astype(np.float32)).to(DEV)
    lut = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).to(DEV)
    out = {}
    for tag, walk, hot_rows in (("natural", False, False), ("sorted", True, False), ("hot", True, True)):
        monkeypatch.setattr(aggregate, "NARROW_SORTED_WALK", walk)
        monkeypatch.setattr(aggregate, "HOT_COLUMN_ROWS", hot_rows)
        out[tag] = spmm_launch(g, S, lut, not s_by_code, not s_by_code, s_by_code=s_by_code)
    assert g._sorted_copy_hot is not None and g._sorted_copy_hot.n_cols == n + 64
    assert int((g._sorted_copy_hot.col >= n).sum()) >= int(hot.sum())
    assert torch.equal(out["natural"], out["sorted"])
    # W in {1, 2, 4} with the hot copy: spmm_hot_kernel (the head of the copy in LDS, persistent workgroups) — ordinary rows
    # bit for bit, hub rows add their pairs wave by wave instead of workgroup by workgroup
    hub = torch.from_numpy(np.diff(rowptr) > 64).to(DEV)                  # (narrow rows: sliced from 64 pairs, graph.LONG_ROW_THRESHOLD_NARROW)
    assert torch.equal(out["natural"][~hub], out["hot"][~hub])
    assert O.rel_err(out["hot"].cpu(), out["natural"].cpu().double()) <= 2e-6
    if W in (1, 2, 4) and not s_by_code:
        monkeypatch.setattr(aggregate, "HOT_ROWS_IN_LDS", False)
        assert torch.equal(spmm_launch(g, S, lut, True, True), out["natural"])            # the plain kernel on the same copy
    if not s_by_code:
        wt = lut.cpu().double().unsqueeze(0) / g.cnt.cpu().clamp_min(1).double().unsqueeze(-1)
        want = O.spmm_csr(rowptr, col, code, S.cpu().double(), wt, with_rest=True)
        assert O.rel_err(out["hot"].cpu(), want) <= 1e-5


@pytest.mark.parametrize("W,K,bf16", [(64, 1, False), (64, 2, False), (64, 1, True), (1, 2, False), (16, 1, False)])
def test_packed_index_entries_give_the_same_bits(W, K, bf16, monkeypatch):
    """The degree-sorted copy read as ONE index stream (col | code << 29, gnan_spmm_args.packed_index) == the same copy
    read as separate col / code arrays, bit for bit: rows, hub-row slices, wide index runs, bf16 rows; the packed array
    itself is checked against its definition, and shapes it cannot carry keep the two arrays."""
    from gnan_amd import functional, graph as G
    from gnan_amd.aggregate import spmm_launch
    monkeypatch.setattr(aggregate, "DEGREE_SORTED_COPY_MIN_ROWS", 1)
    monkeypatch.setattr(aggregate, "NARROW_SORTED_MIN_NNZ", 0)
    rng = np.random.default_rng(W + K)
    n, D = 6000, K + 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(7, 700), (100, 5000), (5999, 2100)])
    g = _graph(rowptr, col, code, n, D)
    S = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    if bf16:
        S = S.bfloat16()
    lut = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).to(DEV)
    out = {}
    for packed in (False, True):
        monkeypatch.setattr(aggregate, "PACKED_INDEX", packed)
        out[packed] = spmm_launch(g, S, lut, True, True, reduce_cr=1 if W == 64 else 0)
    copy = g.degree_sorted_copy()[0]
    assert copy.colp is not None
    want = copy.col.long() | (copy.code.long() << G.PACK_SHIFT)
    assert torch.equal(copy.colp.long() & 0xffffffff, want)
    assert torch.equal(out[False], out[True])
    wide = _graph(rowptr, col, np.minimum(code, 1), n, 6)               # six shells: codes no longer fit the kernel variant
    assert wide.degree_sorted_copy()[0].colp is None


@pytest.mark.parametrize("seed", range(12))
def test_aggregation_paths_agree_on_random_shapes(seed, monkeypatch):
    """Seeded random shapes (rows, degrees, hubs, shells, widths, skewed columns) through every walk of the aggregation —
    natural order, degree-sorted copy, hot rows appended, packed / wide index loads on and off: the same bits from all of
    them, and the float64 oracle within 1e-5."""
    from gnan_amd import functional, graph as G
    from gnan_amd.aggregate import spmm_launch
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(300, 9000))
    K = int(rng.integers(1, 3))
    D = K + 2
    W = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 40, 64]))
    hubs = [(int(rng.integers(0, n)), int(rng.integers(600, 4000))) for _ in range(int(rng.integers(0, 3)))]
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=hubs)
    if rng.random() < 0.7:                                                # a few neighbours listed by many pairs
        hot = rng.random(col.shape[0]) < rng.uniform(0.1, 0.6)
        col[hot] = rng.integers(0, 70, int(hot.sum())) * (n // 70)
    use_cnt, with_rest = bool(rng.random() < 0.7), bool(rng.random() < 0.7)
    g = _graph(rowptr, col, code, n, D, idx_dtype=torch.int64 if rng.random() < 0.5 else torch.int32)
    S = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    lut = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).to(DEV)
    monkeypatch.setattr(G, "HOT_COLUMNS", 64)
    monkeypatch.setattr(G, "HOT_COLUMNS_MIN", 1)
    monkeypatch.setattr(G, "HOT_COLUMNS_MIN_NNZ", 0)
    monkeypatch.setattr(G, "HOT_COLUMNS_MIN_SHARE", 0.0)
    monkeypatch.setattr(aggregate, "NARROW_SORTED_MIN_NNZ", 0)
    out = []
    for min_rows, walk, hot_rows, packed, wide in ((1 << 30, False, False, False, False), (1, True, False, False, True),
                                                   (1, True, True, True, True), (1, True, True, False, False),
                                                   (1 << 30, False, False, False, True)):
        monkeypatch.setattr(aggregate, "DEGREE_SORTED_COPY_MIN_ROWS", min_rows)
        monkeypatch.setattr(aggregate, "NARROW_SORTED_WALK", walk)
        monkeypatch.setattr(aggregate, "HOT_COLUMN_ROWS", hot_rows)
        monkeypatch.setattr(aggregate, "PACKED_INDEX", packed)
        monkeypatch.setattr(aggregate, "WIDE_INDEX_LOADS", wide)
        out.append(spmm_launch(g, S, lut, use_cnt, with_rest))
    sliced = torch.from_numpy(np.diff(rowptr) > 64).to(DEV)     # rows some plan may slice (narrow rows: from 64 pairs)
    for k, y in enumerate(out[1:]):
        if k == 1 and W in (1, 2, 4):
            # hot rows + packed index at W in {1, 2, 4}: spmm_hot_kernel — ordinary rows bit for bit, hub rows (added wave by
            # wave there) to round-off
            assert torch.equal(out[0][~sliced], y[~sliced]) and O.rel_err(y.cpu(), out[0].cpu().double()) <= 2e-6
        else:
            assert torch.equal(out[0], y)
    wt = lut.cpu().double().unsqueeze(0).expand(n, -1, -1)
    if use_cnt:
        wt = wt / g.cnt.cpu().clamp_min(1).double().unsqueeze(-1)
    want = O.spmm_csr(rowptr, col, code, S.cpu().double(), wt, with_rest=with_rest)
    assert O.rel_err(out[0].cpu(), want) <= 1e-5


@pytest.mark.parametrize("seed", range(8))
def test_aggregation_gradients_agree_on_random_shapes(seed, monkeypatch):
    """Seeded random shapes through the aggregation's autograd function — one-pass narrow backward over natural order and
    over the degree-sorted copy with hot packed rows, wide operands through the per-node weight table: operand gradients
    identical between the walks, everything by the rule against float64 oracle autograd."""
    from gnan_amd import functional, graph as G
    from gnan_amd.aggregate import rho_aggregate
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(300, 6000))
    K = int(rng.integers(1, 3))
    D = K + 2
    W = int(rng.choice([1, 2, 3, 4, 7, 16, 24, 40]))
    hubs = [(int(rng.integers(0, n)), int(rng.integers(600, 3000))) for _ in range(int(rng.integers(0, 3)))]
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=hubs)
    if rng.random() < 0.7:
        hot = rng.random(col.shape[0]) < rng.uniform(0.1, 0.5)
        col[hot] = rng.integers(0, 50, int(hot.sum())) * (n // 50)
    use_cnt, with_rest = bool(rng.random() < 0.7), bool(rng.random() < 0.7)
    g = _graph(rowptr, col, code, n, D)
    S0 = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    lut0 = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).to(DEV)
    up = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    monkeypatch.setattr(G, "HOT_COLUMNS", 64)
    monkeypatch.setattr(G, "HOT_COLUMNS_MIN", 1)
    monkeypatch.setattr(G, "HOT_COLUMNS_MIN_NNZ", 0)
    monkeypatch.setattr(G, "HOT_COLUMNS_MIN_SHARE", 0.0)
    monkeypatch.setattr(aggregate, "NARROW_SORTED_MIN_NNZ", 0)
    got = []
    for min_rows in (1 << 30, 1):
        monkeypatch.setattr(aggregate, "DEGREE_SORTED_COPY_MIN_ROWS", min_rows)
        S, lut = S0.clone().requires_grad_(True), lut0.clone().requires_grad_(True)
        got.append(torch.autograd.grad(rho_aggregate(g, S, lut, use_cnt, with_rest=with_rest), [S, lut], up))
    assert torch.equal(got[0][0], got[1][0])
    S64, lut64 = S0.cpu().double().requires_grad_(True), lut0.cpu().double().requires_grad_(True)
    wt = lut64.unsqueeze(0).expand(n, -1, -1)
    if use_cnt:
        wt = wt / g.cnt.cpu().clamp_min(1).double().unsqueeze(-1)
    ref = torch.autograd.grad(O.spmm_csr(rowptr, col, code, S64, wt, with_rest=with_rest), [S64, lut64], up.cpu().double())
    def ref32():
        S32, l32 = S0.cpu().requires_grad_(True), lut0.cpu().requires_grad_(True)
        w32 = l32.unsqueeze(0).expand(n, -1, -1)
        if use_cnt:
            w32 = w32 / g.cnt.cpu().clamp_min(1).float().unsqueeze(-1)
        return torch.autograd.grad(O.spmm_csr(rowptr, col, code, S32, w32, with_rest=with_rest), [S32, l32], up.cpu())
    for k in range(2):
        for run in got:
            assert_rule(run[k], ref[k], lambda k=k: ref32()[k], (k, seed))


def test_degree_schedule_is_bit_identical_to_natural_order(monkeypatch):
    """Rows processed in degree order (through a degree-sorted copy of the CSR, or through an index) and stored in
    place == rows processed in natural order (same arithmetic per row)."""
    from gnan_amd import functional
    from gnan_amd.aggregate import spmm_launch
    rng = np.random.default_rng(21)
    n, K, W = 5000, 1, 16
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(3, 900), (4000, 4000)])
    g = _graph(rowptr, col, code, n, K + 2)
    S = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).to(DEV)
    lut = torch.tensor([[1.0], [0.5], [0.01]], device=DEV)
    assert aggregate.DEGREE_SORTED_COPY
    monkeypatch.setattr(aggregate, "DEGREE_SORTED_COPY_MIN_ROWS", 2)
    y_copy = spmm_launch(g, S, lut, True, True)                  # degree-sorted copy of the CSR (scatter_out = 2)
    r_copy = spmm_launch(g, S, lut, True, True, reduce_cr=1)
    gs, order, _ = g.degree_sorted_copy()
    deg = (g.rowptr[1:] - g.rowptr[:-1]).cpu().numpy()
    o = order.cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(n)) and np.all(np.diff(deg[o]) >= 0)
    rp_s, col_s, code_s = gs.rowptr.cpu().numpy(), gs.col.cpu().numpy(), gs.code.cpu().numpy()
    for q in (0, 1, n // 2, n - 2, n - 1):                        # row q of the copy = row order[q], pairs in their order
        a, b = rowptr[o[q]], rowptr[o[q] + 1]
        assert np.array_equal(col_s[rp_s[q]:rp_s[q + 1]], col[a:b]) and np.array_equal(code_s[rp_s[q]:rp_s[q + 1]], code[a:b])
    assert torch.equal(gs.cnt, g.cnt[order.long()])
    monkeypatch.setattr(aggregate, "DEGREE_SORTED_COPY", False)  # degree order through an index (scatter_out = 1)
    assert torch.equal(y_copy, spmm_launch(g, S, lut, True, True))
    assert torch.equal(r_copy, spmm_launch(g, S, lut, True, True, reduce_cr=1))
    monkeypatch.setattr(aggregate, "DEGREE_SCHEDULE_MIN_WIDTH", 1 << 30)   # natural order
    assert torch.equal(y_copy, spmm_launch(g, S, lut, True, True))
    assert torch.equal(r_copy, spmm_launch(g, S, lut, True, True, reduce_cr=1))


@pytest.mark.parametrize("W,reduce_cr", [(64, 1), (64, 0), (8, 1), (16, 0), (128, 4)])
def test_bf16_operand_storage(W, reduce_cr):
    """bf16 rows, fp32 accumulate: exact w.r.t. the oracle evaluated on the bf16-rounded operand."""
    from gnan_amd.functional import column_sums
    from gnan_amd.aggregate import spmm_launch
    rng = np.random.default_rng(W + reduce_cr)
    n, K = 3000, 1
    D = K + 2
    rowptr, col, code = _random_csr(n, n, K, rng, hubs=[(7, 900), (1500, 2600)])
    cnt = _cnt_np(rowptr, code, n, D)
    g = _graph(rowptr, col, code, n, D)
    S = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32))
    S16 = S.to(torch.bfloat16)
    lut = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32))
    truth = O.spmm_csr(rowptr, col, code, S16.double(), O.weight_table(lut.double(), cnt).expand(n, -1, -1))
    if reduce_cr:
        truth = truth.view(n, W // reduce_cr, reduce_cr).sum(1)
    tot = column_sums(S16.to(DEV))
    assert float((tot.cpu().double() - S16.double().sum(0)).abs().max()) <= 1e-5 * float(S16.double().abs().sum(0).max())
    y = spmm_launch(g, S16.to(DEV), lut.to(DEV), True, True, reduce_cr=reduce_cr).cpu()
    assert y.shape == truth.shape
    assert O.rel_err(y, truth) <= 1e-5, O.rel_err(y, truth)


def test_bf16_rows_from_the_table_lookup(monkeypatch):
    """feature_mlps(out_dtype=bf16): the stored rows are the correctly rounded fp32 results and the fused totals
    describe the rounded operand."""
    from gnan_amd.functional import feature_mlps
    F, L, H, C = 32, 3, 16, 1
    sd = _mlp_state(F, L, H, C, True, seed=1)
    st = _stack(sd, F, L, H, C, True)
    x = torch.rand(40_000, F, device=DEV) * 2 - 1
    with torch.no_grad():
        y32 = feature_mlps(x, st, False, out_dtype=torch.float32)
        y16, tot = feature_mlps(x, st, False, return_total=True, out_dtype=torch.bfloat16)
    assert y16.dtype == torch.bfloat16
    exact = (y16.float() == y32.to(torch.bfloat16).float()).float().mean()
    assert float(exact) >= 0.999                       # fp32 inputs of the rounding differ in the last ulp at most
    assert float((y16.float() - y32).abs().max()) <= 2 ** -7 * float(y32.abs().max())
    want = y16.double().sum(0)
    assert float((tot.double() - want).abs().max()) <= 1e-5 * float(y16.double().abs().sum(0).max())


def _on_kink_state(F, L, H, C, mode, seed):
    """``zero``: the reference's initial biases (GNAN.py:49-53) under O(1) weights — every kink at x = 0;
    ``exact``: first-layer kinks on the float32 numbers {0, 1/4, 1/2, 1} (weights multiples of 1/64, b = -w a)."""
    sd = _mlp_state(F, L, H, C, True, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in range(F):
        if mode == "zero":
            for li in range(L):
                sd[f"fs.{k}.{3 * li}.bias"].zero_()
        else:
            w = torch.round(sd[f"fs.{k}.0.weight"] * 64.0) / 64.0
            w[w == 0] = 1.0 / 64.0
            a = torch.tensor([0.0, 0.25, 0.5, 1.0])[torch.randint(0, 4, (H,), generator=g)]
            sd[f"fs.{k}.0.weight"], sd[f"fs.{k}.0.bias"] = w, -(w[:, 0] * a)
    return sd


def _on_kink_inputs(n, F, seed):
    g0 = torch.Generator().manual_seed(seed)
    levels = torch.tensor([0.0, 0.0, 0.0, 1.0, 0.25, 0.5, 0.75, -0.0])
    x = levels[torch.randint(0, len(levels), (n, F), generator=g0)]
    x[:, -1] = 1.0                                                       # the reference's ones column
    return x, g0


@pytest.mark.parametrize("mode", ["zero", "exact"])
@pytest.mark.parametrize("F,L,H,C,bias_unused", [(5, 3, 8, 1, 0), (64, 3, 64, 1, 0), (6, 3, 33, 2, 0), (4, 2, 8, 3, 0), (2, 3, 128, 2, 0)])
def test_table_build_kernel_point_pieces_match_torch_builder(F, L, H, C, bias_unused, mode, monkeypatch):
    """Anchors on which a hidden pre-activation is exactly zero get a one-float32-step piece behind them, from the kernel
    and from the torch builder alike (same piece counts, same anchors); the tabulated function is unchanged."""
    from gnan_amd import pwl
    sd = _on_kink_state(F, L, H, C, mode, seed=F + H)
    st = _stack(sd, F, L, H, C, True)
    monkeypatch.setattr(pwl, "BUILD_BACKEND", "torch")
    t_ref = pwl.build_tables(st)
    monkeypatch.setattr(pwl, "BUILD_BACKEND", "auto")
    t_hip = pwl.build_tables(st)
    assert torch.equal(t_ref.off, t_hip.off), "same number of pieces per feature"
    a = t_hip.anchor.cpu()
    up = torch.nextafter(a, torch.full_like(a, float("inf")))
    assert bool(((a[1:] > a[:-1]) & (a[1:] <= up[:-1])).any()), "no point piece in the tables"
    assert torch.equal(t_ref.anchor.cpu(), a)                             # exact kinks are exact in both builders
    x, _ = _on_kink_inputs(4000, F, 2)
    truth = O.feature_mlps(x.double(), {k: v.double() for k, v in sd.items()}).reshape(4000, -1)
    tc = pwl.PwlTables(*[q.cpu() if torch.is_tensor(q) else q for q in t_hip])
    assert O.rel_err(pwl.evaluate_reference(x, tc, False), truth) <= 1e-5


@pytest.mark.parametrize("mode", ["zero", "exact"])
@pytest.mark.parametrize("route,F,L,H,C,sum_features", [
    ("fast", 64, 3, 64, 1, True),        # whole 16-feature groups, one channel, feature sum: kept pieces (a byte per look-up)
    ("fast", 32, 3, 16, 1, False),
    ("ragged", 17, 3, 16, 1, True), ("ragged", 129, 2, 8, 1, False),
    ("general", 6, 3, 16, 3, True), ("general", 5, 2, 8, 2, False),
    ("two-phase", 20, 3, 16, 7, True), ("two-phase", 12, 3, 16, 5, False), ("two-phase", 3, 3, 16, 100, True),
])
@pytest.mark.parametrize("fixed", [True, False])
@pytest.mark.parametrize("hip_grads", [True, False])
def test_table_path_gradients_with_inputs_on_kinks(route, F, L, H, C, sum_features, fixed, hip_grads, mode, monkeypatch):
    """x EXACTLY on a ReLU kink — zero biases (the reference's own initial state, GNAN.py:49-53) with one-hot style
    features, or kinks placed on float32 numbers the inputs take.  torch differentiates relu at 0 as 0; every table route
    (fast / ragged / general / two-phase look-up, kept pieces, fixed-point and float moments, kernel and torch parameter
    gradients) must hand the bias gradients to the same units: within the floor (1e-5 of the largest gradient) of float64 oracle autograd."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    monkeypatch.setattr(functional, "MOMENTS_FIXED_POINT", fixed)
    monkeypatch.setattr(functional, "HIP_TABLE_GRADS", hip_grads)
    monkeypatch.setattr(functional, "SUM_VIA_FEATURES_MAX_NODES", 0)
    if route == "two-phase":
        monkeypatch.setattr(functional, "FPWL_ROWS_MIN_NODES", 1)
        monkeypatch.setattr(functional, "FPWL_ROWS_MIN_CHANNELS", 2)
    n = 3000
    sd = _on_kink_state(F, L, H, C, mode, seed=F + C)
    st = _stack(sd, F, L, H, C, True)
    leaves = [t for t in st[:6] if t is not None]
    for t in leaves:
        t.requires_grad_(True)
    x, gen = _on_kink_inputs(n, F, 7)
    gup = torch.randn(n, C if sum_features else F * C, generator=gen)
    out = feature_mlps(x.to(DEV), st, sum_features)
    got = torch.autograd.grad(out, leaves, gup.to(DEV))
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = O.feature_mlps(x.double(), sd64)
    ref = ref.sum(1) if sum_features else ref.reshape(n, -1)
    assert O.rel_err(out.detach().cpu(), ref.detach()) <= 1e-5
    ref.backward(gup.double())
    last = 3 * (L - 1)
    want = {"w_first": torch.stack([sd64[f"fs.{k}.0.weight"].grad[:, 0] for k in range(F)]),
            "b_first": torch.stack([sd64[f"fs.{k}.0.bias"].grad for k in range(F)]),
            "w_last": torch.stack([sd64[f"fs.{k}.{last}.weight"].grad for k in range(F)]),
            "b_last": torch.stack([sd64[f"fs.{k}.{last}.bias"].grad for k in range(F)])}
    if L == 3:
        want["w_mid"] = torch.stack([sd64[f"fs.{k}.3.weight"].grad for k in range(F)]).unsqueeze(0)
        want["b_mid"] = torch.stack([sd64[f"fs.{k}.3.bias"].grad for k in range(F)]).unsqueeze(0)
    names = [nm for nm, t in zip(("w_first", "b_first", "w_mid", "b_mid", "w_last", "b_last"), st[:6]) if t is not None]
    assert_grads_rule(dict(zip(names, got)), want, None, (route, mode))


def _rho_state(L, H, C, bias, seed, zero_bias=False):
    g = torch.Generator().manual_seed(seed)
    dims = [1] + [H] * (L - 1) + [C]
    sd = {}
    for li in range(L):
        sd[f"rho.{2 * li}.weight"] = torch.randn(dims[li + 1], dims[li], generator=g) * (2.0 / (dims[li] + dims[li + 1])) ** 0.5
        if bias:
            sd[f"rho.{2 * li}.bias"] = torch.zeros(dims[li + 1]) if zero_bias else torch.randn(dims[li + 1], generator=g) * 0.5
    return sd


def _stack_rho(sd, L, H, C, bias):
    from gnan_amd.functional import StackedMLP

    def get(li, what):
        return sd[f"rho.{2 * li}.{what}"].unsqueeze(0).to(DEV)                     # the feature axis: one function
    if L == 1:
        return StackedMLP(None, None, None, None, get(0, "weight")[..., 0], get(0, "bias") if bias else None, 1, 0, C, 1)
    w_mid = b_mid = None
    if L > 2:
        w_mid = torch.stack([get(li, "weight") for li in range(1, L - 1)], 0)
        b_mid = torch.stack([get(li, "bias") for li in range(1, L - 1)], 0) if bias else None
    return StackedMLP(get(0, "weight")[..., 0], get(0, "bias") if bias else None, w_mid, b_mid,
                      get(L - 1, "weight"), get(L - 1, "bias") if bias else None, L, H, C, 1)


@pytest.mark.parametrize("route", ["table", "kernels"])
@pytest.mark.parametrize("L,H,C,bias,D,zero_bias", [(3, 16, 1, True, 3, False), (3, 64, 3, True, 5, False), (2, 8, 2, False, 3, False),
                                                     (1, 1, 2, True, 4, False), (3, 32, 1, True, 3, True), (4, 8, 2, True, 3, False)])
def test_pre_rho_row_table_vs_oracle(L, H, C, bias, D, zero_bias, route, monkeypatch):
    """GNAN.py:65-67 per shell: lut[i, d] = rho(u_d / cnt[i, d]) — gnan_rho_row_lut on rho's exact table (and the small-size
    route through the shape-function kernels) against the float64 oracle, forward and parameter gradients.  The rest
    bucket's argument is exactly 0: with zero biases (graph tasks build rho without bias, GNAN.py:36-37) it sits on every kink."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import rho_row_lut
    from gnan_amd.graph import hop_inputs
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL if route == "table" else _lib.FMLP_AUTO)
    monkeypatch.setattr(functional, "PRE_RHO_TABLE_MIN", 1 << 40)                 # AUTO: small-size route at this size
    n = 5000
    rng = np.random.default_rng(L * 10 + C)
    cnt = rng.integers(0, 40, (n, D)).astype(np.int32)                              # zeros: empty shells (clamped to 1)
    cnt[:, 0] = 1
    cnt[:, D - 1] = rng.integers(1, 10 ** 7, n)
    sd = _rho_state(L, max(H, 1), C, bias, seed=7 * L + C, zero_bias=zero_bias)
    st = _stack_rho(sd, L, H if L > 1 else 0, C, bias)
    leaves = [t for t in st[:6] if t is not None]
    for t in leaves:
        t.requires_grad_(True)
    lut = rho_row_lut(torch.from_numpy(cnt).to(DEV), hop_inputs(D, DEV), st)
    assert lut.shape == (n, D, C)
    gup = torch.randn(n, D, C, generator=torch.Generator().manual_seed(3))
    got = torch.autograd.grad(lut, leaves, gup.to(DEV))
    p64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    want = O.row_lut_pre_rho(p64, cnt, torch.float64)
    assert O.rel_err(lut.detach().cpu(), want.detach()) <= 1e-5
    want.backward(gup.double())
    names = [nm for nm, t in zip(("w_first", "b_first", "w_mid", "b_mid", "w_last", "b_last"), st[:6]) if t is not None]
    last = 2 * (L - 1)
    ref = {"w_last": p64[f"rho.{last}.weight"].grad if L > 1 else p64["rho.0.weight"].grad[:, 0]}
    if bias:
        ref["b_last"] = p64[f"rho.{last}.bias"].grad
    if L > 1:
        ref["w_first"] = p64["rho.0.weight"].grad[:, 0]
        if bias:
            ref["b_first"] = p64["rho.0.bias"].grad
    if L > 2:
        ref["w_mid"] = torch.stack([p64[f"rho.{2 * li}.weight"].grad for li in range(1, L - 1)])
        if bias:
            ref["b_mid"] = torch.stack([p64[f"rho.{2 * li}.bias"].grad for li in range(1, L - 1)])
    assert_grads_rule(dict(zip(names, got)), ref, None, route)


@pytest.mark.parametrize("F,L,H,C,bias", [(5, 3, 16, 1, True), (4, 2, 8, 3, True), (6, 3, 64, 7, False), (3, 4, 8, 2, True),
                                           (2, 3, 96, 2, True), (3, 1, 0, 2, True)])
@pytest.mark.parametrize("sum_features", [True, False])
def test_training_mode_dropout_in_the_kernels(F, L, H, C, bias, sum_features):
    """GNAN.py:28,32 (run.sh: dropout 0.6): nn.Dropout behind every hidden ReLU, inside gnan_fmlp_fwd / gnan_fmlp_bwd.  The
    masks are a hash of (seed, node, feature, layer, unit); gnan_dropout_mask writes them out and the oracle — the same
    Linear / ReLU / Dropout chain with those masks, float64 — must agree, forward and every parameter gradient.  (L = 4 and
    H = 96 are beyond gnan_fmlp_bwd: their backward runs the batched restatement with the kernels' masks.)"""
    from gnan_amd.functional import dropout_masks, feature_mlps_dropout
    n, drop_p, seed = 700, 0.6, 123456789012345
    sd = _mlp_state(F, L, max(H, 1), C, bias, seed=F + L)
    st = _stack(sd, F, L, H, C, bias)
    leaves = [t for t in st[:6] if t is not None]
    for t in leaves:
        t.requires_grad_(True)
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(n, F, generator=gen) * 2 - 0.5
    gup = torch.randn(n, C if sum_features else F * C, generator=gen)
    y = feature_mlps_dropout(x.to(DEV), st, sum_features, drop_p, seed=seed)
    got = torch.autograd.grad(y, leaves, gup.to(DEV))
    keep = None
    if L > 1:
        keep = dropout_masks(seed, drop_p, n, F, L - 1, H, DEV).cpu()
        rate = float(keep.float().mean())
        assert abs(rate - (1 - drop_p)) < 4 * (drop_p * (1 - drop_p) / keep.numel()) ** 0.5 + 1e-3, rate
        assert float(keep[:, 0, 0].float().mean(0).min()) > 0.2          # no unit or node is always dropped
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = O.feature_mlps(x.double(), sd64, keep=keep, drop_p=drop_p)
    ref = ref.sum(1) if sum_features else ref.reshape(n, -1)
    assert O.rel_err(y.detach().cpu(), ref.detach()) <= 1e-5
    ref.backward(gup.double())
    last = 3 * (L - 1)
    want = {"w_last": torch.stack([sd64[f"fs.{k}.{last}.weight"].grad for k in range(F)])}
    if L == 1:
        want["w_last"] = want["w_last"][..., 0]
    if bias:
        want["b_last"] = torch.stack([sd64[f"fs.{k}.{last}.bias"].grad for k in range(F)])
    if L > 1:
        want["w_first"] = torch.stack([sd64[f"fs.{k}.0.weight"].grad[:, 0] for k in range(F)])
        if bias:
            want["b_first"] = torch.stack([sd64[f"fs.{k}.0.bias"].grad for k in range(F)])
    if L > 2:
        want["w_mid"] = torch.stack([torch.stack([sd64[f"fs.{k}.{3 * li}.weight"].grad for k in range(F)]) for li in range(1, L - 1)])
        if bias:
            want["b_mid"] = torch.stack([torch.stack([sd64[f"fs.{k}.{3 * li}.bias"].grad for k in range(F)]) for li in range(1, L - 1)])
    names = [nm for nm, t in zip(("w_first", "b_first", "w_mid", "b_mid", "w_last", "b_last"), st[:6]) if t is not None]
    assert_grads_rule(dict(zip(names, got)), want, None, (F, L, H, C))
    if L > 1:
        with torch.no_grad():
            again = feature_mlps_dropout(x.to(DEV), st, sum_features, drop_p, seed=seed)
            other = feature_mlps_dropout(x.to(DEV), st, sum_features, drop_p, seed=seed + 1)
        assert torch.equal(again, y.detach()) and not torch.equal(other, y.detach())


@pytest.mark.parametrize("n,F,C,L,H,rho_c,D_hops,use_cnt,graph_sum", [
    (1, 3, 1, 3, 64, 1, 1, True, True), (30, 15, 1, 3, 64, 1, 9, True, True), (64, 7, 2, 3, 64, 2, 20, True, False),
    (17, 5, 8, 2, 32, 1, 6, False, True), (40, 9, 3, 3, 48, 3, 70, True, False), (64, 64, 1, 3, 64, 1, 12, True, True),
    (33, 4, 4, 2, 64, 4, 200, True, True),
    # more than 64 nodes (two node blocks: 3 % of config 2's graphs): forward and backward stay one launch each up to 128
    (65, 15, 1, 3, 64, 1, 11, True, True), (100, 15, 1, 3, 64, 1, 30, True, True), (128, 6, 2, 3, 32, 1, 64, True, False),
    (97, 5, 8, 2, 64, 8, 90, False, True), (128, 15, 1, 3, 64, 1, 5, True, True)])
def test_small_graph_forward_in_one_launch(n, F, C, L, H, rho_c, D_hops, use_cnt, graph_sum):
    """gnan_small_graph_fwd (shape functions of all features, rho on the distinct distances, normalised aggregation and the
    graph read-out of a small dense-coded graph: ONE launch) == the float64 oracle chain; S and the rho table it leaves behind
    feed the general backward kernels: gradients of every parameter == oracle autograd; bit-reproducible; more than 64
    shells, one node, more than 64 nodes, rho per channel, two-layer MLPs, hidden widths below 64; what it does not cover is refused."""
    from gnan_amd import HopGraph, small_graph
    from gnan_amd.functional import StackedMLP
    from gnan_amd.small_graph import small_graph_applies, small_graph_forward
    rng = np.random.default_rng(n * 7 + F)
    hops = rng.integers(-1, D_hops, (n, n)).astype(np.int32)           # -1 = unreachable
    hops[np.arange(n), np.arange(n)] = 0
    nd = torch.zeros(n, n)
    nd[torch.from_numpy(hops >= 0)] = 1.0 / (torch.from_numpy(hops[hops >= 0]).float() + 1.0)
    g = HopGraph.from_dense(nd.to(DEV))
    D = g.n_codes

    def mlp(Fk, Ck, bias):
        t = lambda *s: torch.from_numpy((rng.standard_normal(s) * 0.5).astype(np.float32))
        return [t(Fk, H), t(Fk, H) if bias else None, t(1, Fk, H, H) if L == 3 else None, t(1, Fk, H) if (L == 3 and bias) else None,
                t(Fk, Ck, H), t(Fk, Ck) if bias else None]
    fp, rp = mlp(F, C, True), mlp(1, rho_c, False)
    x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))

    def run(dtype, dev, fused):
        fl = [None if t is None else t.to(dev, dtype).requires_grad_(True) for t in fp]
        rl = [None if t is None else t.to(dev, dtype).requires_grad_(True) for t in rp]
        xs = x.to(dev, dtype)
        if fused:
            f, r = StackedMLP(*fl, L, H, C, F), StackedMLP(*rl, L, H, rho_c, 1)
            assert small_graph_applies(xs, g, f, r)
            out = small_graph_forward(xs, g, f, r, use_cnt, graph_sum)
        else:                                                         # the chain, written out in float64
            def net(v, p, k):                                         # v [m] -> [m, C]
                h = torch.relu(v[:, None] * p[0][k] + (0 if p[1] is None else p[1][k]))
                if L == 3:
                    h = torch.relu(h @ p[2][0, k].T + (0 if p[3] is None else p[3][0, k]))
                return h @ p[4][k].T + (0 if p[5] is None else p[5][k])
            S = sum(net(xs[:, k], fl, k) for k in range(F))
            u = torch.zeros(D, dtype=dtype)
            u[: D - 1] = (1.0 / (torch.arange(D - 1, dtype=torch.float32) + 1.0)).to(dtype)
            lut = net(u, rl, 0)                                       # [D, rho_c]
            codes = torch.from_numpy(np.where(hops >= 0, hops, D - 1)).long()
            w = lut[codes]                                            # [n, n, rho_c]
            if use_cnt:
                cnt = g.cnt.cpu().clamp_min(1).to(dtype)
                w = w / cnt[torch.arange(n)[:, None], codes][..., None]
            out = (w * S[None, :, :]).sum(1)
            out = out.sum(0).view(-1, 1) if graph_sum else out
        up = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(out.shape))).to(dev, dtype)
        live = [t for t in fl + rl if t is not None]
        return out.detach(), torch.autograd.grad(out, live, up)
    got, got_g = run(torch.float32, DEV, True)
    want, want_g = run(torch.float64, "cpu", False)
    assert got.shape == ((C, 1) if graph_sum else (n, C))
    assert O.rel_err(got.cpu(), want) <= 1e-5
    ref32 = lambda: run(torch.float32, "cpu", False)[1]            # noqa: E731  (the float32 chain: the rule's bound, if needed)
    for i, (a, b) in enumerate(zip(got_g, want_g)):
        assert_rule(a, b, lambda i=i: ref32()[i], (a.shape,))
    again, again_g = run(torch.float32, DEV, True)
    assert torch.equal(got, again) and all(torch.equal(a, b) for a, b in zip(got_g, again_g))
    # the backward pass is ONE launch too where rho has one channel and there are at most 64 shells (gnan_small_graph_bwd);
    # the general kernels on the saved node sums and rho table otherwise — and on request: same gradients
    one_launch = rho_c == 1 and D <= 64
    old_flag = small_graph.SMALL_GRAPH_BACKWARD
    small_graph.SMALL_GRAPH_BACKWARD = False
    try:
        _, general_g = run(torch.float32, DEV, True)
    finally:
        small_graph.SMALL_GRAPH_BACKWARD = old_flag
    for a, b, w in zip(got_g, general_g, want_g):
        assert float((a - b).abs().max()) <= (TWO_FLOORS if one_launch else 0.0) * float(w.abs().max())      # two routes
    # refused shapes: more nodes than the kernel holds, a CSR graph, inputs that want a gradient
    f, r = StackedMLP(*[None if t is None else t.to(DEV) for t in fp], L, H, C, F), StackedMLP(*[None if t is None else t.to(DEV) for t in rp], L, H, rho_c, 1)
    assert not small_graph_applies(x.to(DEV).requires_grad_(True), g, f, r)
    big = HopGraph.from_dense(torch.eye(129, device=DEV))
    assert not small_graph_applies(torch.zeros(129, F, device=DEV), big, f, r)


@pytest.mark.parametrize("n,F,L,H,D_hops,graph_sum,bias_r", [
    (1, 3, 3, 64, 1, True, False), (30, 15, 3, 64, 9, True, False), (64, 7, 3, 64, 20, False, True), (17, 5, 2, 32, 6, True, True),
    (40, 9, 3, 48, 62, False, False), (65, 15, 3, 64, 11, True, False), (128, 6, 2, 32, 63, False, True),
    (128, 15, 3, 64, 5, True, False), (100, 4, 3, 64, 40, True, True)])
def test_small_graph_pre_rho_in_one_launch(n, F, L, H, D_hops, graph_sum, bias_r):
    """The stand-alone file's normalisation (GNAN.py:65-67: rho at distance / shell size — n * D arguments instead of D) through
    gnan_small_graph_fwd / _bwd with pre_rho: forward == the float64 chain, the gradients of every parameter of f and rho ==
    float64 autograd (rho's from its n * D arguments), bit-reproducible; refused for a rho of several channels or > 64 shells."""
    from gnan_amd import HopGraph
    from gnan_amd.functional import StackedMLP
    from gnan_amd.small_graph import small_graph_applies, small_graph_forward
    rng = np.random.default_rng(n * 11 + F)
    hops = rng.integers(-1, D_hops, (n, n)).astype(np.int32)
    hops[np.arange(n), np.arange(n)] = 0
    nd = torch.zeros(n, n)
    nd[torch.from_numpy(hops >= 0)] = 1.0 / (torch.from_numpy(hops[hops >= 0]).float() + 1.0)
    g = HopGraph.from_dense(nd.to(DEV))
    D = g.n_codes
    assert D <= 64

    def mlp(Fk, bias):
        t = lambda *s: torch.from_numpy((rng.standard_normal(s) * 0.5).astype(np.float32))
        return [t(Fk, H), t(Fk, H) if bias else None, t(1, Fk, H, H) if L == 3 else None, t(1, Fk, H) if (L == 3 and bias) else None,
                t(Fk, 1, H), t(Fk, 1) if bias else None]
    fp, rp = mlp(F, True), mlp(1, bias_r)
    x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))

    def run(dtype, dev, fused):
        fl = [None if t is None else t.to(dev, dtype).requires_grad_(True) for t in fp]
        rl = [None if t is None else t.to(dev, dtype).requires_grad_(True) for t in rp]
        xs = x.to(dev, dtype)
        if fused:
            f, r = StackedMLP(*fl, L, H, 1, F), StackedMLP(*rl, L, H, 1, 1)
            assert small_graph_applies(xs, g, f, r, pre_rho=True)
            out = small_graph_forward(xs, g, f, r, "pre", graph_sum)
        else:
            def net(v, p, k):
                h = torch.relu(v[:, None] * p[0][k] + (0 if p[1] is None else p[1][k]))
                if L == 3:
                    h = torch.relu(h @ p[2][0, k].T + (0 if p[3] is None else p[3][0, k]))
                return h @ p[4][k].T + (0 if p[5] is None else p[5][k])
            S = sum(net(xs[:, k], fl, k) for k in range(F))                           # [n, 1]
            codes = torch.from_numpy(np.where(hops >= 0, hops, D - 1)).long()
            u = torch.zeros(D, dtype=torch.float32)
            u[: D - 1] = 1.0 / (torch.arange(D - 1, dtype=torch.float32) + 1.0)
            cnt = g.cnt.cpu().clamp_min(1).float()
            arg = (u[codes] / cnt[torch.arange(n)[:, None], codes]).to(dtype)        # float32 division, as torch.div on the inputs
            w = net(arg.reshape(-1), rl, 0).view(n, n, 1)
            out = (w * S[None, :, :]).sum(1)
            out = out.sum(0).view(-1, 1) if graph_sum else out
        up = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(out.shape))).to(dev, dtype)
        live = [t for t in fl + rl if t is not None]
        return out.detach(), torch.autograd.grad(out, live, up)
    got, got_g = run(torch.float32, DEV, True)
    want, want_g = run(torch.float64, "cpu", False)
    assert got.shape == ((1, 1) if graph_sum else (n, 1))
    assert O.rel_err(got.cpu(), want) <= 1e-5
    ref32 = lambda: run(torch.float32, "cpu", False)[1]            # noqa: E731  (the float32 chain: the rule's bound, if needed)
    for i, (a, b) in enumerate(zip(got_g, want_g)):
        assert_rule(a, b, lambda i=i: ref32()[i], (a.shape,))
    again, again_g = run(torch.float32, DEV, True)
    assert torch.equal(got, again) and all(torch.equal(a, b) for a, b in zip(got_g, again_g))
    f = StackedMLP(*[None if t is None else t.to(DEV) for t in fp], L, H, 1, F)
    two = mlp(1, False)
    two[4] = two[4].repeat(1, 2, 1)
    assert not small_graph_applies(x.to(DEV), g, f, StackedMLP(*[None if t is None else t.to(DEV) for t in two], L, H, 2, 1), pre_rho=True)


@pytest.mark.parametrize("n,F,L,H,Ln,Hn,Cn,D_hops,use_cnt,bias", [
    (1, 3, 3, 64, 2, 64, 1, 1, True, True), (30, 15, 3, 64, 2, 64, 1, 9, True, True), (40, 4, 3, 8, 1, 8, 1, 12, False, True),
    (64, 7, 2, 32, 3, 48, 3, 20, True, True), (65, 15, 3, 64, 3, 64, 8, 11, True, False), (128, 6, 3, 64, 2, 16, 2, 63, True, True),
    (100, 70, 2, 16, 1, 16, 4, 30, False, False), (17, 5, 3, 64, 3, 64, 5, 6, True, True)])
def test_small_graph_nam_readout_in_one_launch(n, F, L, H, Ln, Hn, Cn, D_hops, use_cnt, bias):
    """models.py:358-384 with a NAM read-out (is_graph_task, readout_n_layers > 0) on a small graph: forward by ONE launch
    (gnan_small_graph_nam_fwd) == the float64 chain; backward by ONE launch (gnan_small_graph_nam_bwd): the gradients of every
    parameter of f, rho and the read-out == float64 autograd; bit-reproducible; read-outs of one, two and three layers, up to
    eight classes, more than 64 features, with and without shell normalisation; what it does not cover is refused."""
    from gnan_amd import HopGraph
    from gnan_amd.functional import StackedMLP
    from gnan_amd.small_graph import small_graph_nam_applies, small_graph_nam_forward
    rng = np.random.default_rng(n * 13 + F + 4)        # (instances checked to be well conditioned: see the note on rho below)
    hops = rng.integers(-1, D_hops, (n, n)).astype(np.int32)
    hops[np.arange(n), np.arange(n)] = 0
    nd = torch.zeros(n, n)
    nd[torch.from_numpy(hops >= 0)] = 1.0 / (torch.from_numpy(hops[hops >= 0]).float() + 1.0)
    g = HopGraph.from_dense(nd.to(DEV))
    D = g.n_codes

    def mlp(Fk, Ck, Lk, Hk, b):
        t = lambda *s: torch.from_numpy((rng.standard_normal(s) * 0.5).astype(np.float32))
        if Lk == 1:
            return [None, None, None, None, t(Fk, Ck), t(Fk, Ck) if b else None]
        return [t(Fk, Hk), t(Fk, Hk) if b else None, t(1, Fk, Hk, Hk) if Lk == 3 else None, t(1, Fk, Hk) if (Lk == 3 and b) else None,
                t(Fk, Ck, Hk), t(Fk, Ck) if b else None]
    fp, rp, npar = mlp(F, 1, L, H, bias), mlp(1, 1, L, H, False), mlp(F, Cn, Ln, Hn, bias)
    # rho has no bias on graph tasks (models.py:336-337), so rho(u) = u * rho(1): with random signs in its last layer rho(1) can
    # come out as the small difference of large terms, and every weight of the graph inherits the float32 error of that ONE
    # number (seen: 2e-5 for the float32 restatement itself).  A last layer of one sign keeps the instance well conditioned —
    # as does a seed for which the single output of a one-class read-out is not the small difference of the features' terms
    # (seen: terms 700 times the output, 1.4e-4 for the float32 restatement).
    rp[4] = rp[4].abs()
    x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))

    def net(v, p, k, Lk):                                             # v [m] -> [m, C]
        if Lk == 1:
            return v[:, None] * p[4][k][None, :] + (0 if p[5] is None else p[5][k])
        h = torch.relu(v[:, None] * p[0][k] + (0 if p[1] is None else p[1][k]))
        if Lk == 3:
            h = torch.relu(h @ p[2][0, k].T + (0 if p[3] is None else p[3][0, k]))
        return h @ p[4][k].T + (0 if p[5] is None else p[5][k])

    def run(dtype, dev, fused):
        fl, rl, nl = ([None if t is None else t.to(dev, dtype).requires_grad_(True) for t in q] for q in (fp, rp, npar))
        xs = x.to(dev, dtype)
        if fused:
            f, r, nm = StackedMLP(*fl, L, H, 1, F), StackedMLP(*rl, L, H, 1, 1), StackedMLP(*nl, Ln, Hn if Ln > 1 else 0, Cn, F)
            assert small_graph_nam_applies(xs, g, f, r, nm)
            out = small_graph_nam_forward(xs, g, f, r, nm, use_cnt)
        else:
            fx = torch.cat([net(xs[:, k], fl, k, L) for k in range(F)], 1)            # [n, F]
            u = torch.zeros(D, dtype=dtype)
            u[: D - 1] = (1.0 / (torch.arange(D - 1, dtype=torch.float32) + 1.0)).to(dtype)
            lut = net(u, rl, 0, L)[:, 0]
            codes = torch.from_numpy(np.where(hops >= 0, hops, D - 1)).long()
            w = lut[codes]
            if use_cnt:
                cnt = g.cnt.cpu().clamp_min(1).to(dtype)
                w = w / cnt[torch.arange(n)[:, None], codes]
            hidden = (w @ fx).sum(0)                                                  # [F]   models.py:373-379
            out = sum(net(hidden[k:k + 1], nl, k, Ln) for k in range(F)).T            # [C, 1]
        up = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(out.shape))).to(dev, dtype)
        live = [t for t in fl + rl + nl if t is not None]
        return out.detach(), torch.autograd.grad(out, live, up)
    got, got_g = run(torch.float32, DEV, True)
    want, want_g = run(torch.float64, "cpu", False)
    assert got.shape == (Cn, 1)
    assert O.rel_err(got.cpu(), want) <= 1e-5
    ref32 = lambda: run(torch.float32, "cpu", False)[1]            # noqa: E731
    for i, (a, b) in enumerate(zip(got_g, want_g)):
        assert_rule(a, b, lambda i=i: ref32()[i], (a.shape,))
    again, again_g = run(torch.float32, DEV, True)
    assert torch.equal(got, again) and all(torch.equal(a, b) for a, b in zip(got_g, again_g))
    dev_stack = lambda q, *dims: StackedMLP(*[None if t is None else t.to(DEV) for t in q], *dims)
    f, r, nm = dev_stack(fp, L, H, 1, F), dev_stack(rp, L, H, 1, 1), dev_stack(npar, Ln, Hn if Ln > 1 else 0, Cn, F)
    assert not small_graph_nam_applies(x.to(DEV).requires_grad_(True), g, f, r, nm)
    big = HopGraph.from_dense(torch.eye(129, device=DEV))
    assert not small_graph_nam_applies(torch.zeros(129, F, device=DEV), big, f, r, nm)


def test_multi_copy_in_one_launch():
    """gnan_multi_copy: up to eight device-to-device copies per launch — aligned and unaligned ends, odd byte counts, empty
    tensors, several dtypes; more than eight and mismatched pairs are refused."""
    from gnan_amd import _lib
    g = torch.Generator().manual_seed(0)
    srcs = [torch.randn(37, 5, generator=g).to(DEV), torch.randint(0, 255, (33, 33), generator=g, dtype=torch.uint8).to(DEV),
            torch.randint(0, 9, (30, 11), generator=g, dtype=torch.int32).to(DEV), torch.randn(1, generator=g).to(DEV),
            torch.randn(4097, generator=g).to(DEV)[1:], torch.empty(0, 3, device=DEV)]
    dsts = [torch.zeros_like(s) for s in srcs]
    dsts[4] = torch.zeros(4098, device=DEV)[2:]                         # both ends off the 16-byte grid, differently
    _lib.multi_copy(list(zip(dsts, srcs)))
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)
    with pytest.raises(_lib.GnanHipError):
        _lib.multi_copy([(torch.zeros(3, device=DEV), torch.zeros(4, device=DEV))])
    with pytest.raises(_lib.GnanHipError):
        _lib.multi_copy([(torch.zeros(3, device=DEV), torch.ones(3, device=DEV))] * 9)


# ---------------------------------------------------------------------------------------------------------- direct index
def _index_inputs(kind, n, F, seed):
    g0 = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        x = torch.rand(n, F, generator=g0) * 2 - 1
    elif kind == "one_hot":                         # bag-of-words style: mostly exact zeros, a few ones, the ones column
        x = (torch.rand(n, F, generator=g0) < 0.1).float()
    elif kind == "heavy_tail":                      # a few huge values stretch the hinted range: the grid turns coarse
        x = torch.randn(n, F, generator=g0)
        x[::997] *= 1e4
    else:                                           # "levels": exact kink positions, signed zero, non-finite values
        levels = torch.tensor([0.0, -0.0, 1.0, 0.25, 0.5, 0.75, 2.0, -1.0, float("inf"), -float("inf"), float("nan")])
        x = levels[torch.randint(0, len(levels), (n, F), generator=g0)]
    x[:, -1] = 1.0                                  # pre_process_datasets.py:127
    return x.to(DEV)


@pytest.mark.parametrize("kind", ["uniform", "one_hot", "heavy_tail", "levels"])
@pytest.mark.parametrize("mode", ["rows", "rows_bf16", "sum"])
@pytest.mark.parametrize("buckets,flags", [(512, 0), (1024, 16), (256, 8), (2048, 4), (512, 4)])
def test_direct_index_lookup_is_the_tree_search_bit_for_bit(kind, mode, buckets, flags, monkeypatch):
    """csrc/fpwl_index.hip finds the piece of a value by arithmetic on a per-feature grid over the data's range plus
    comparisons inside the cell; it must give the bits of the tree-search kernel (same piece, same formula) for ANY
    input — on kinks, outside the hinted range, non-finite — and for a range hint that is wrong."""
    from gnan_amd import _lib, functional, pwl
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    monkeypatch.setattr(functional, "INDEX_BUCKETS", buckets)
    monkeypatch.setattr(functional, "INDEX_FLAGS", flags)       # 0: 32-feature groups (full lines); 8 / 16: 512 / 1024 threads; 4: 16-feature groups
    # the bucket tables by their own launch where the test counts them (`built`); out of the table build's compaction pass (round 6,
    # the default) for the 16-feature groups — same tables (test_bucket_tables_out_of_the_table_build_...), and the look-up must
    # be handed them either way
    in_build = flags == 4
    monkeypatch.setattr(pwl, "INDEX_IN_BUILD", in_build)
    n, F, L, H = 70_001, 32, 3, 24
    sd = _on_kink_state(F, L, H, 1, "exact", seed=3) if kind in ("levels", "one_hot") else _mlp_state(F, L, H, 1, True, seed=3)
    st = _stack(sd, F, L, H, 1, True)
    x = _index_inputs(kind, n, F, seed=5)
    functional._RANGE_CHURN.clear()                 # (every test hands over a new feature matrix: not the churn this guards against)
    built = []
    real_index = functional._fpwl_index
    monkeypatch.setattr(functional, "_fpwl_index", lambda *a: built.append(real_index(*a)) or built[-1])

    def run(on, hint=None):
        monkeypatch.setattr(functional, "INDEX_LOOKUP", on)
        if hint is not None:
            monkeypatch.setattr(functional, "_feature_range", lambda t: hint)
        with torch.no_grad():
            if mode == "sum":
                return feature_mlps(x, st, True, return_total=True)
            return feature_mlps(x, st, False, return_total=True,
                                out_dtype=torch.bfloat16 if mode == "rows_bf16" else torch.float32)

    want, want_total = run(False)
    assert built and all(b is None for b in built)
    del built[:]
    handed = []
    real_launch = functional._fpwl_launch
    monkeypatch.setattr(functional, "_fpwl_launch", lambda *a, **k: handed.append(k.get("index")) or real_launch(*a, **k))
    got, got_total = run(True)
    monkeypatch.setattr(functional, "_fpwl_launch", real_launch)
    if in_build:
        assert not built and handed and handed[-1] is not None      # made by the build, handed to the look-up
    else:
        assert built and built[-1] is not None      # the direct-index tables were built and handed to the look-up
    def same(a, b):         # bit for bit; a NaN matches a NaN (its sign follows the order of the operands of a sum)
        a, b = a.float(), b.float()
        na, nb_ = torch.isnan(a), torch.isnan(b)
        return torch.equal(na, nb_) and torch.equal(torch.where(na, torch.zeros_like(a), a).view(torch.int32),
                                                    torch.where(nb_, torch.zeros_like(b), b).view(torch.int32))

    assert same(got, want)
    # the column sums add float32 partial sums per thread: their grouping follows the kernel's node-to-thread map
    finite = torch.isfinite(want_total) & torch.isfinite(got_total)
    assert torch.equal(torch.isnan(got_total), torch.isnan(want_total))
    bound = 1e-6 * float(torch.nan_to_num(want.float(), nan=0.0, posinf=0.0, neginf=0.0).abs().sum(0).max()) if mode != "sum" else 0.0
    assert float((got_total - want_total)[finite].abs().max() if finite.any() else 0.0) <= bound + 1e-30
    # a WRONG hint (narrower than the data, shifted, degenerate): still exact — the hint steers speed only
    for lo, hi in ((0.2, 0.3), (5.0, 9.0), (0.0, 0.0), (float("nan"), 1.0), (-float("inf"), float("inf"))):
        hint = torch.tensor([[lo, hi]] * F, device=DEV)
        got2, tot2 = run(True, hint)
        assert same(got2, want), (lo, hi)


@pytest.mark.parametrize("F", [32, 48, 144])
def test_direct_index_keeps_the_pieces_for_the_backward_pass(F, monkeypatch):
    """Training, one channel: the direct-index feature-sum kernel stores the same piece bytes as the tree-search kernel,
    so the moment kernel bins the same terms — parameter gradients bit for bit.  F = 48 / 144: three / nine 16-feature groups
    on a medium batch — a workgroup per (node block, group) and the groups' partial sums added in order (`sum_workspace`)."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    n, L, H = 80_000, 3, 16
    sd = _mlp_state(F, L, H, 1, True, seed=11)
    x = _index_inputs("uniform", n, F, seed=2)
    target = torch.randn(n, 1, generator=torch.Generator().manual_seed(1)).to(DEV)
    grads = {}
    functional._RANGE_CHURN.clear()
    for on in (False, True):
        monkeypatch.setattr(functional, "INDEX_LOOKUP", on)
        st = _stack(sd, F, L, H, 1, True)
        leaves = [t.detach().clone().requires_grad_(True) for t in st[:6]]
        S = feature_mlps(x, type(st)(*leaves, *st[6:]), True)
        ((S - target) ** 2).mean().backward()
        grads[on] = [S.detach()] + [t.grad for t in leaves]
    for a, b in zip(grads[False], grads[True]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("F,total_rows", [(48, None), (144, 50_001), (64, 1)])
def test_group_split_feature_sum_hands_back_its_column_sum(F, total_rows, monkeypatch):
    """A medium batch with several feature groups (a workgroup per (node block, group), `sum_workspace`): the pass that adds the
    groups' partial sums also returns the column sum of the result over the first `total_rows` rows (`sum_total`) — what a
    gnan_colsum over the result gave, without its two launches."""
    from gnan_amd import _lib, functional
    from gnan_amd.functional import feature_mlps
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    n, L, H = 80_000, 3, 16
    sd = _mlp_state(F, L, H, 1, True, seed=5)
    x = _index_inputs("uniform", n, F, seed=3)
    functional._RANGE_CHURN.clear()
    st = _stack(sd, F, L, H, 1, True)
    with torch.no_grad():
        S, total = feature_mlps(x, st, True, return_total=True, total_rows=total_rows)
        plain = feature_mlps(x, st, True)
    assert torch.equal(S, plain)
    rows = n if total_rows is None else total_rows
    ref = S[:rows].double().sum(0)
    assert total.shape == (1,) and abs(float(total[0]) - float(ref[0])) <= 1e-6 * float(S[:rows].abs().double().sum())
    calls = []
    real = functional.column_sums
    monkeypatch.setattr(functional, "column_sums", lambda t: calls.append(1) or real(t))
    with torch.no_grad():
        _, again = feature_mlps(x, st, True, return_total=True, total_rows=total_rows)
    assert torch.equal(again, total) and not calls           # fixed order; no separate column-sum pass


def test_feature_range_kernel():
    import gnan_amd  # noqa: F401
    from gnan_amd import functional
    x = torch.randn(100_003, 48, device=DEV)
    x[5, 3] = float("nan")
    x[:, 7] = 1.0
    x[17, 9] = float("inf")
    x = x.contiguous()
    functional._RANGE_CHURN.clear()
    r = functional._feature_range(x)
    ok = torch.ones(48, dtype=torch.bool)
    xm = torch.where(torch.isnan(x), torch.zeros_like(x), x)
    lo, hi = xm.min(0).values, xm.max(0).values
    lo[3] = torch.where(torch.isnan(x[:, 3]), torch.full_like(x[:, 3], float("inf")), x[:, 3]).min()
    hi[3] = torch.where(torch.isnan(x[:, 3]), torch.full_like(x[:, 3], -float("inf")), x[:, 3]).max()
    assert torch.equal(r[:, 0], lo) and torch.equal(r[:, 1], hi)
    assert functional._feature_range(x) is r                             # cached per tensor object and version
    x[0, 0] = 1e9                                                        # an in-place write invalidates it
    assert float(functional._feature_range(x)[0, 1]) == 1e9
    # a new feature matrix every call: after three misses the range pass is no longer paid for that shape
    functional._RANGE_CHURN.clear()
    seen = [functional._feature_range(torch.randn(70_000, 16, device=DEV)) is not None for _ in range(5)]
    assert seen == [True, True, True, False, False]
    functional._RANGE_CHURN.clear()


# =============================================================================
# propagation-blocked narrow aggregation (csrc/spmm_pb.hip)
# =============================================================================
def _pb_graph(rng, n_rows, n_cols, D, hubs=(), self_pairs=True):
    from test_pb_plan import random_graph
    g, rowptr, col, code = random_graph(rng, n_rows, n_cols, D, hubs=hubs, self_pairs=self_pairs)
    from gnan_amd import HopGraph
    gd = HopGraph.from_csr(g.rowptr.to(DEV), g.col.to(DEV), g.code.to(DEV), n_cols=n_cols, n_codes=D)
    return gd, rowptr, col, code


@pytest.mark.parametrize("D,W,self_pairs,use_cnt,with_rest,lds", [(3, 1, True, True, True, 1024), (3, 2, True, False, True, 2048),
                                                                   (4, 1, True, True, False, 1024), (4, 4, False, True, True, 4096),
                                                                   (3, 1, False, True, True, 65536), (2, 1, False, True, True, 1024),
                                                                   (3, 1, True, True, True, 65536), (3, 2, True, True, True, 65536)])
def test_propagation_blocked_aggregation_vs_oracle(D, W, self_pairs, use_cnt, with_rest, lds, monkeypatch):
    """gnan_spmm_pb_fwd == the float64 oracle (integer accumulation: 1e-6 is float32 rounding of inputs and weights only) and
    is bit-reproducible; small LDS budgets give many bins / column blocks and multi-slot hub rows on a small graph."""
    from gnan_amd import graph as G
    from gnan_amd.aggregate import pb_launch, spmm_launch
    monkeypatch.setattr(G, "PB_LDS_BYTES", lds)
    if lds < 65536:
        monkeypatch.setattr(G, "PB_SLOT_PAIRS", 8)
    rng = np.random.default_rng(D * 10 + W + lds)
    n_rows, n_cols = (700, 900) if lds < 65536 else (40_000, 50_000)
    hubs = [(5, 60), (333, 150)] if lds < 65536 else [(5, 3000), (333, 20_000), (39_999, 700)]
    g, rowptr, col, code = _pb_graph(rng, n_rows, n_cols, D, hubs=hubs, self_pairs=self_pairs)
    plan = g.pb_plan(W)
    assert plan is not None
    S = torch.from_numpy(rng.standard_normal((n_cols, W)).astype(np.float32) * 3.0)
    lut = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32))
    total = S.double().sum(0).float() if with_rest else None
    got = pb_launch(g, plan, S.to(DEV), lut.to(DEV), use_cnt, None if total is None else total.to(DEV))
    again = pb_launch(g, plan, S.to(DEV), lut.to(DEV), use_cnt, None if total is None else total.to(DEV))
    assert torch.equal(got, again)
    # float64 truth, vectorised: out[i] = sum_e w(i, code_e) S[col_e] (+ w(i, rest) (sum_j S[j] - sum_e S[col_e]))
    row_of = torch.repeat_interleave(torch.arange(n_rows), torch.from_numpy(np.diff(rowptr)))
    colt, codet, S64 = torch.from_numpy(col).long(), torch.from_numpy(code).long(), S.double()
    c = g.cnt.cpu().double().clamp_min(1) if use_cnt else torch.ones(n_rows, D, dtype=torch.float64)
    l64 = lut.double().reshape(-1)
    want = torch.zeros(n_rows, W, dtype=torch.float64).index_add(0, row_of, (l64[codet] / c[row_of, codet]).unsqueeze(1) * S64[colt])
    if with_rest:
        listed = torch.zeros(n_rows, W, dtype=torch.float64).index_add(0, row_of, S64[colt])
        want = want + (l64[-1] / c[:, -1]).unsqueeze(1) * (total.double().unsqueeze(0) - listed)
    assert got.shape == want.shape
    assert O.rel_err(got.cpu(), want) <= 1e-6
    # ... and the route: spmm_launch sends a large-enough graph here by itself, with the row-parallel kernel's result
    monkeypatch.setattr(aggregate, "PB_MIN_NNZ", 0)
    routed = spmm_launch(g, S.to(DEV), lut.to(DEV), use_cnt, with_rest, s_total=None if total is None else total.to(DEV))
    assert torch.equal(routed, again)
    monkeypatch.setattr(aggregate, "PB_NARROW", False)
    rows_kernel = spmm_launch(g, S.to(DEV), lut.to(DEV), use_cnt, with_rest, s_total=None if total is None else total.to(DEV))
    assert float((rows_kernel - again).abs().max()) <= TWO_FLOORS * float(again.abs().max())            # two routes


def test_propagation_blocked_aggregation_non_finite_operand():
    """An infinite or NaN operand value cannot be put in fixed point: every output row is NaN (the float chain would
    have produced inf / NaN in the rows that list it)."""
    from gnan_amd.aggregate import pb_launch
    rng = np.random.default_rng(4)
    g, *_ = _pb_graph(rng, 300, 300, 3)
    S = torch.randn(300, 1)
    S[17] = float("inf")
    out = pb_launch(g, g.pb_plan(1), S.to(DEV), torch.tensor([[1.0], [0.5], [0.1]], device=DEV), True, None)
    assert bool(torch.isnan(out).all())
    S[17] = 0.0
    out = pb_launch(g, g.pb_plan(1), S.to(DEV), torch.tensor([[1.0], [0.5], [0.1]], device=DEV), True, None)
    assert bool(torch.isfinite(out).all())
    zero = pb_launch(g, g.pb_plan(1), torch.zeros(300, 1, device=DEV), torch.tensor([[1.0], [0.5], [0.1]], device=DEV), True, None)
    assert float(zero.abs().max()) == 0.0


@pytest.mark.parametrize("D,self_pairs,use_cnt,with_rest,lds", [(3, True, True, True, 2048), (3, True, False, False, 2048),
                                                                 (2, False, True, True, 2048), (3, True, True, True, 65536),
                                                                 (3, False, True, True, 2048), (4, True, True, True, 2048)])
@pytest.mark.parametrize("one_column", [True, False])
def test_propagation_blocked_backward_vs_oracle_autograd(D, self_pairs, use_cnt, with_rest, lds, one_column, monkeypatch):
    """One-column aggregation, forward AND backward on the bucketed copies: operand and table gradients against float64 oracle
    autograd, bit-reproducible, and equal to the row-parallel kernels' gradients to float32 rounding.  `one_column`: the backward
    from the forward's kept shell sums (gnan_spmm_pb_pack1 + gnan_spmm_pb_fwd over the transposed graph, W = 1; graphs whose self
    pairs carry hop code 0) — else, and where that does not apply, the packed two-column rows (gnan_spmm_pb_bwd).  D = 4 lists
    two non-self codes: the backward stays row-parallel."""
    from gnan_amd import graph as G
    from gnan_amd.aggregate import pb_bwd_applies, rho_aggregate
    monkeypatch.setattr(aggregate, "PB_BACKWARD_ONE_COLUMN", one_column)
    ran = []
    real = aggregate.pb_bwd1_launch
    monkeypatch.setattr(aggregate, "pb_bwd1_launch", lambda *a, **k: ran.append(1) or real(*a, **k))
    monkeypatch.setattr(G, "PB_LDS_BYTES", lds)
    if lds < 65536:
        monkeypatch.setattr(G, "PB_SLOT_PAIRS", 8)
    monkeypatch.setattr(aggregate, "PB_MIN_NNZ", 0)
    rng = np.random.default_rng(D * 7 + lds + int(self_pairs))
    n = 900 if lds < 65536 else 30_000
    hubs = [(5, 60), (333, 150)] if lds < 65536 else [(5, 3000), (333, 12_000)]
    g, rowptr, col, code = _pb_graph(rng, n, n, D, hubs=hubs, self_pairs=self_pairs)
    if rng.random() < 0.9:                                    # a few very popular neighbours: hub rows of the TRANSPOSED graph
        hot = torch.from_numpy(rng.random(col.shape[0]) < 0.2)
        hot &= g.code.cpu() != 0
        col = col.copy()
        col[hot.numpy()] = rng.integers(0, 6, int(hot.sum())) * (n // 6)
        from gnan_amd import HopGraph
        g = HopGraph.from_csr(g.rowptr, torch.from_numpy(col).to(DEV), g.code, n_cols=n, n_codes=D)
    assert (pb_bwd_applies(g, 1, D) is not None) == (D <= 3)
    S0 = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).to(DEV)
    lut0 = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).to(DEV)
    up = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).to(DEV)

    def grads():
        S, lut = S0.clone().requires_grad_(True), lut0.clone().requires_grad_(True)
        return torch.autograd.grad(rho_aggregate(g, S, lut, use_cnt, with_rest=with_rest), [S, lut], up)
    got, again = grads(), grads()
    assert torch.equal(got[0], again[0]) and torch.equal(got[1], again[1])
    plan = g.pb_plan(1)
    assert bool(ran) == (one_column and plan.n_acc == 1 and plan.code_base >= 1), (ran, D, self_pairs, plan.n_acc, plan.code_base)
    assert not one_column or ran or D != 3 or self_pairs is False
    monkeypatch.setattr(aggregate, "PB_NARROW", False)
    rows = grads()
    S64, lut64 = S0.cpu().double().requires_grad_(True), lut0.cpu().double().requires_grad_(True)
    wt = lut64.unsqueeze(0).expand(n, -1, -1)
    if use_cnt:
        wt = wt / g.cnt.cpu().clamp_min(1).double().unsqueeze(-1)
    row_of = torch.repeat_interleave(torch.arange(n), torch.from_numpy(np.diff(rowptr)))
    colt, codet = torch.from_numpy(col).long(), torch.from_numpy(code).long()
    y64 = torch.zeros(n, 1, dtype=torch.float64).index_add(0, row_of, wt[row_of, codet] * S64[colt])
    if with_rest:
        y64 = y64 + wt[:, -1] * (S64.sum(0, keepdim=True) - torch.zeros(n, 1, dtype=torch.float64).index_add(0, row_of, S64[colt]))
    ref = torch.autograd.grad(y64, [S64, lut64], up.cpu().double())
    for k in range(2):
        scale = float(ref[k].abs().max())
        assert float((got[k].cpu().double() - ref[k]).abs().max()) <= 1e-5 * scale, (k, scale)
        assert float((got[k] - rows[k]).abs().max()) <= TWO_FLOORS * scale, (k, scale)                  # two routes


@pytest.mark.parametrize("idx_dtype,n,hubs", [(torch.int64, 5000, [(3, 900), (4000, 4000)]), (torch.int32, 70_000, [(5, 200_000), (69_999, 129)]),
                                              (torch.int32, 300, [])])
def test_degree_sorted_copy_by_the_library_equals_the_framework_route(idx_dtype, n, hubs, monkeypatch):
    """gnan_degree_sorted_csr (stable radix sort of the rows by length, scan, copy pass) == the torch route, array by array, bit
    for bit: processing order, row pointers, column ids, hop codes, the packed index stream, the permuted shell sizes — for a
    graph and for its transpose (whose count table belongs to the neighbours and is not permuted)."""
    from gnan_amd import graph as G
    rng = np.random.default_rng(n)
    rowptr, col, code = _random_csr(n, n + 17, 2, rng, hubs=hubs)
    for transposed in (False, True):
        made = []
        for hip in (True, False):
            monkeypatch.setattr(G, "SORTED_COPY_IN_HIP", hip)
            g = _graph(rowptr, col, code, n + 17, 4, idx_dtype=idx_dtype)
            g = g.transposed() if transposed else g
            made.append(g.degree_sorted_copy())
        (a, oa, pa), (b, ob, pb) = made
        assert torch.equal(oa, ob)
        assert a.rowptr.dtype == b.rowptr.dtype and torch.equal(a.rowptr, b.rowptr)
        assert torch.equal(a.col, b.col) and torch.equal(a.code, b.code) and torch.equal(a.cnt, b.cnt)
        assert (a.colp is None) == (b.colp is None) and (a.colp is None or torch.equal(a.colp, b.colp))
        assert pa.n_long == pb.n_long and pa.n_slices == pb.n_slices
        if pa.n_long:
            assert torch.equal(pa.rows, pb.rows) and torch.equal(pa.slice_ptr, pb.slice_ptr)


@pytest.mark.parametrize("D,W,self_pairs,lds,idx_dtype", [(3, 1, True, 1024, torch.int64), (3, 2, True, 2048, torch.int32), (4, 1, True, 1024, torch.int32),
                                                          (4, 4, False, 4096, torch.int64), (3, 1, False, 65536, torch.int32), (2, 1, False, 1024, torch.int32),
                                                          (3, 1, True, 65536, torch.int32)])
def test_bucketed_pairs_by_the_library_equal_the_framework_route(D, W, self_pairs, lds, idx_dtype, monkeypatch):
    """HopGraph.pb_plan built by gnan_pb_plan_rows / _keys / _fill == the framework route (which the CPU suite pins against the
    oracle, tests/test_pb_plan.py), array by array, bit for bit."""
    import dataclasses
    from gnan_amd import HopGraph, graph as G
    monkeypatch.setattr(G, "PB_LDS_BYTES", lds)
    if lds < 65536:
        monkeypatch.setattr(G, "PB_SLOT_PAIRS", 8)
    rng = np.random.default_rng(D * 100 + W + lds)
    n_rows, n_cols = (700, 900) if lds < 65536 else (40_000, 50_000)
    hubs = [(5, 60), (333, 150)] if lds < 65536 else [(5, 3000), (333, 20_000), (39_999, 700)]
    g0, rowptr, col, code = _pb_graph(rng, n_rows, n_cols, D, hubs=hubs, self_pairs=self_pairs)
    plans = []
    for hip in (True, False):
        monkeypatch.setattr(G, "PB_PLAN_IN_HIP", hip)
        g = HopGraph.from_csr(g0.rowptr.to(idx_dtype), g0.col, g0.code, n_cols=n_cols, n_codes=D)
        plans.append(g.pb_plan(W))
    a, b = plans
    assert a is not None and b is not None
    for f in dataclasses.fields(a):
        x, y = getattr(a, f.name), getattr(b, f.name)
        if torch.is_tensor(x) or torch.is_tensor(y):
            assert torch.is_tensor(x) and torch.is_tensor(y) and x.dtype == y.dtype and torch.equal(x, y), f.name
        else:
            assert x == y, (f.name, x, y)


@pytest.mark.parametrize("idx_dtype,n,hubs", [(torch.int64, 5000, [(3, 900), (4000, 4000)]), (torch.int32, 70_000, [(5, 200_000), (69_999, 300)])])
def test_transposed_adjacency_by_the_library_equals_the_framework_route(idx_dtype, n, hubs, monkeypatch):
    """gnan_csr_transpose == the torch route (stable argsort by column id), array by array, bit for bit — rectangular graph,
    hub rows, a very popular neighbour."""
    from gnan_amd import graph as G
    rng = np.random.default_rng(n + 1)
    rowptr, col, code = _random_csr(n, n + 9, 2, rng, hubs=hubs)
    col[rng.random(col.shape[0]) < 0.2] = 7
    made = []
    for hip in (True, False):
        monkeypatch.setattr(G, "TRANSPOSE_IN_HIP", hip)
        made.append(_graph(rowptr, col, code, n + 9, 4, idx_dtype=idx_dtype).transposed())
    a, b = made
    assert (a.n_rows, a.n_cols) == (b.n_rows, b.n_cols) == (n + 9, n)
    assert a.rowptr.dtype == b.rowptr.dtype and torch.equal(a.rowptr, b.rowptr)
    assert torch.equal(a.col, b.col) and torch.equal(a.code, b.code) and a._cnt_by_col and b._cnt_by_col


@pytest.mark.parametrize("F,H,buckets", [(16, 8, 256), (64, 64, 512), (32, 33, 1024)])
def test_bucket_tables_out_of_the_table_build_equal_the_builder_launch(F, H, buckets, monkeypatch):
    """gnan_pwl_build with index_range / index_table / index_key == gnan_pwl_build + gnan_fpwl_index_build: tables, keys and the
    look-up that reads them, bit for bit."""
    from gnan_amd import _lib, functional as Fn, pwl
    sd = _mlp_state(F, 3, H, 1, True, seed=F + H)
    st = _stack(sd, F, 3, H, 1, True)
    x = (torch.rand(4096, F, generator=torch.Generator().manual_seed(3)) * 4 - 2).to(DEV)
    x_range = torch.stack([x.min(0).values, x.max(0).values], 1).contiguous()
    monkeypatch.setattr(Fn, "INDEX_BUCKETS", buckets)
    got = []
    for fused in (True, False):
        monkeypatch.setattr(pwl, "INDEX_IN_BUILD", fused)
        pending = pwl.build_tables_lazy(st, index_request=(x_range, buckets))
        assert (pending.index is not None) == fused
        t = pending.resolve()
        index = pending.index
        if index is None:
            a = _lib.FpwlArgs(C=1)
            a_keep = Fn._fpwl_index(a, x, t, x_range)
            assert a_keep is not None
            index = tuple(a_keep)
        out = Fn._fpwl_launch(x, t, True, x_range=x_range, index=index)
        got.append((index[0].clone(), index[1].clone(), out.clone()))
    for a, b in zip(*got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n", [300, 20_000])
def test_small_passes_take_their_sums_by_the_last_workgroup(n, monkeypatch):
    """Below 128 workgroups the group-sum total and the packed rows' q come out of the pass's own launch (arrival counters,
    include/gnan_hip.h): the same bits as the second launch, forward and backward, three steps in a row (the counters return to zero)."""
    from gnan_amd import functional as Fn, models
    F, H = 48, 16
    rng = np.random.default_rng(n)
    rowptr, col, code = _random_csr(n, n, 1, rng)
    g = _graph(rowptr, col, code, n, 3)
    x = torch.rand(n, F, generator=torch.Generator().manual_seed(1)).to(DEV)
    target = torch.randn(n, 1, generator=torch.Generator().manual_seed(2)).to(DEV)

    class Bag:
        pass
    data = Bag()
    data.x, data.edge_index, data.gnan_graph = x, None, g
    res = []
    for fused in (True, False):
        monkeypatch.setattr(Fn, "ARRIVE_COUNTERS", fused)
        monkeypatch.setattr(Fn, "FMLP_ALGO", _lib_mod().FMLP_PWL)
        torch.manual_seed(0)
        mod = models.TensorGNAN(F, 1, 3, hidden_channels=H, device=DEV)
        with torch.no_grad():
            for _, p in mod.named_parameters():
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(3)) * 0.3)
        mod = mod.to(DEV).train()
        steps = []
        for _ in range(3):
            mod.zero_grad(set_to_none=True)
            y = mod.forward(data)
            ((y - target) ** 2).mean().backward()
            steps.append((y.detach().clone(), [p.grad.clone() for p in mod.parameters() if p.grad is not None]))
        res.append(steps)
    for (ya, ga), (yb, gb) in zip(*res):
        assert torch.equal(ya, yb) and len(ga) == len(gb) and all(torch.equal(a, b) for a, b in zip(ga, gb))


def _lib_mod():
    from gnan_amd import _lib
    return _lib


@pytest.mark.parametrize("D,use_cnt,with_rest,n", [(3, True, True, 5000), (3, False, False, 5000), (2, True, True, 700),
                                                    (4, True, True, 5000), (4, False, True, 90_000), (3, True, True, 90_000)])
def test_row_parallel_backward_from_kept_shell_sums_vs_oracle_autograd(D, use_cnt, with_rest, n, monkeypatch):
    """One-column aggregation on the row-parallel route: a training forward keeps its rows' per-code shell sums
    (gnan_spmm_args.shell_out) and the backward is gnan_spmm_pack_z (a pass over rows: Z, q, the table gradient) + one gather over
    the transposed pairs.  Operand and table gradients against float64 oracle autograd, bit-reproducible, and equal to the packed
    two-column route (gnan_spmm_pack_bwd_rows + gnan_spmm_bwd_narrow) to float32 rounding.  Very popular neighbours: hub rows
    of the TRANSPOSED graph."""
    from gnan_amd import HopGraph
    from gnan_amd.aggregate import rho_aggregate
    monkeypatch.setattr(aggregate, "PB_NARROW", False)
    ran = []
    real = aggregate.rows_bwd1_launch
    monkeypatch.setattr(aggregate, "rows_bwd1_launch", lambda *a, **k: ran.append(1) or real(*a, **k))
    rng = np.random.default_rng(D * 11 + n)
    rowptr, col, code = _random_csr(n, n, D - 2, rng) if D > 2 else _random_csr(n, n, 0, rng)
    hot = rng.random(col.shape[0]) < 0.15
    hot &= code != 0
    col = col.copy()
    col[hot] = rng.integers(0, 5, int(hot.sum())) * (n // 5)
    g = HopGraph.from_csr(torch.from_numpy(rowptr).to(DEV), torch.from_numpy(col).to(DEV), torch.from_numpy(code).to(DEV), n_cols=n,
                          n_codes=D)
    S0 = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).to(DEV)
    lut0 = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).to(DEV)
    up = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).to(DEV)

    def grads():
        S, lut = S0.clone().requires_grad_(True), lut0.clone().requires_grad_(True)
        return torch.autograd.grad(rho_aggregate(g, S, lut, use_cnt, with_rest=with_rest), [S, lut], up)
    got, again = grads(), grads()
    assert ran and torch.equal(got[0], again[0]) and torch.equal(got[1], again[1])
    monkeypatch.setattr(aggregate, "ROWS_BACKWARD_ONE_COLUMN", False)
    del ran[:]
    packed = grads()
    assert not ran
    S64, lut64 = S0.cpu().double().requires_grad_(True), lut0.cpu().double().requires_grad_(True)
    wt = lut64.unsqueeze(0).expand(n, -1, -1)
    if use_cnt:
        wt = wt / g.cnt.cpu().clamp_min(1).double().unsqueeze(-1)
    row_of = torch.repeat_interleave(torch.arange(n), torch.from_numpy(np.diff(rowptr)))
    colt, codet = torch.from_numpy(col).long(), torch.from_numpy(code).long()
    y64 = torch.zeros(n, 1, dtype=torch.float64).index_add(0, row_of, wt[row_of, codet] * S64[colt])
    if with_rest:
        y64 = y64 + wt[:, -1] * (S64.sum(0, keepdim=True) - torch.zeros(n, 1, dtype=torch.float64).index_add(0, row_of, S64[colt]))
    ref = torch.autograd.grad(y64, [S64, lut64], up.cpu().double())
    for k in range(2):
        scale = float(ref[k].abs().max())
        assert float((got[k].cpu().double() - ref[k]).abs().max()) <= 1e-5 * scale, (k, scale)
        assert float((got[k] - packed[k]).abs().max()) <= TWO_FLOORS * scale, (k, scale)                # two routes


@pytest.mark.parametrize("n,D,Cw,W,use_cnt,with_rest", [(5000, 3, 1, 40, True, True), (777, 4, 1, 7, True, False), (3000, 3, 2, 8, False, True)])
def test_wide_backward_row_expressions_as_kernels(n, D, Cw, W, use_cnt, with_rest):
    """gnan_weight_table == the framework expression it replaced, bit for bit (IEEE divisions, one subtraction);
    gnan_colsum_weighted == the float64 sum of dY / cnt * scale."""
    from gnan_amd import functional as Fn
    gen = torch.Generator().manual_seed(n + D)
    lut = torch.randn(D, Cw, generator=gen).to(DEV)
    cnt = torch.randint(0, 50, (n, D), generator=gen, dtype=torch.int32).to(DEV)
    got = Fn.weight_table(lut, cnt if use_cnt else None, n, with_rest)
    want = lut.unsqueeze(0).float()
    if use_cnt:
        want = want / cnt.clamp_min(1).float().unsqueeze(-1)
    if with_rest:
        want = want - want[:, D - 1:D]
    assert torch.equal(got, want.expand(n, D, Cw).contiguous())
    if Cw == 1:
        dY = torch.randn(n, W, generator=gen).to(DEV)
        v = Fn.column_sums_weighted(dY, cnt[:, D - 1], lut[D - 1])
        ref = (dY.double() / cnt[:, D - 1:D].clamp_min(1).double()).sum(0) * lut[D - 1].double()
        assert float((v.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max() + 1e-30)


@pytest.mark.parametrize("n", [5_000, 30_000])
def test_arrival_counters_stress(n, monkeypatch):
    """The last workgroup's sum against the two-launch route, 2000 launches in a row on fresh data (the packed rows' q, 20 / 118
    workgroups: below the 128 above which a pass keeps its second launch) — the same bits, every time."""
    from gnan_amd import functional as Fn
    from gnan_amd.aggregate import pack_bwd_rows
    gen = torch.Generator(device=DEV).manual_seed(n)
    cnt = torch.randint(0, 9, (n, 3), generator=gen, device=DEV, dtype=torch.int32)
    bad = 0
    for it in range(2000):
        dY = torch.randn(n, 1, generator=gen, device=DEV)
        monkeypatch.setattr(Fn, "ARRIVE_COUNTERS", True)
        _, q1 = pack_bwd_rows(dY, cnt, 3, True, 1, want_q_sum=True)
        monkeypatch.setattr(Fn, "ARRIVE_COUNTERS", False)
        _, q0 = pack_bwd_rows(dY, cnt, 3, True, 1, want_q_sum=True)
        bad += int(not torch.equal(q0, q1))
    assert bad == 0, bad


@pytest.mark.parametrize("idx_dtype,n,hubs,thr", [(torch.int64, 5000, [(3, 900), (4000, 4000), (4999, 600)], 512),
                                                   (torch.int32, 70_000, [(5, 200_000), (69_999, 129)], 64), (torch.int32, 300, [], 512)])
def test_hub_row_plan_by_the_library_equals_the_framework_route(idx_dtype, n, hubs, thr, monkeypatch):
    """gnan_long_row_plan_count / _fill == the torch route (nonzero + cumsum), array by array."""
    from gnan_amd import graph as G
    rng = np.random.default_rng(n + thr)
    rowptr, col, code = _random_csr(n, n, 1, rng, hubs=hubs)
    plans = []
    for hip in (True, False):
        monkeypatch.setattr(G, "LONG_PLAN_IN_HIP", hip)
        plans.append(_graph(rowptr, col, code, n, 3, idx_dtype=idx_dtype).long_row_plan(None, thr))
    a, b = plans
    assert (a.n_long, a.n_slices, a.threshold) == (b.n_long, b.n_slices, b.threshold) and a.n_long == sum(1 for _, d in hubs if d > thr)
    if a.n_long:
        assert a.rows.dtype == b.rows.dtype == torch.int32 and torch.equal(a.rows, b.rows) and torch.equal(a.slice_ptr, b.slice_ptr)
