"""BASELINE.json's full size (R-MAT 10M nodes / 100M edges, 64 feature columns) on the GPU: the oracle cannot run
there, so parity is carried by size-independent properties plus an oracle check on sampled rows.

* sampled rows: float64 CPU restatement of `out[i] = sum_e w[code_e] S[col_e] + w_rest (total - sum_e S[col_e])` on ~250
  random rows (the two largest hub rows and rows just over the hub threshold included) whose operand rows are fetched from the device;
* the reference's evaluation order (aggregate 64 columns, then sum: models.py:373-376) == sum-first (GNAN.py:157) to
  fp32 round-off, through the whole pipeline (tables, look-up, fused read-out / narrow aggregation);
* adjointness  <y, A s> == <A^T y, s>  (forward kernel on the graph vs on its transpose: the backward's dS path);
* linearity in the operand; row subsets are bit-identical to the same rows of the full result."""
import numpy as np
import pytest
import torch

from oracle import gnan_oracle as O
from test_gpu_kernels import _mlp_state, _stack

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, E, F = 10_000_000, 100_000_000, 64


@pytest.fixture(scope="module")
def c4():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    import gnan_amd  # noqa: F401
    from gnan_amd import synthetic as syn
    src, dst = syn.rmat_edges(24, N, E, seed=0, device=DEV)
    g = syn.hop1_csr(src, dst, N)
    del src, dst
    x = syn.block_features(N, F, 0, N, seed=1, device=DEV)
    sd = _mlp_state(F, 3, 64, 1, True, seed=5)
    lut = torch.tensor([[0.9], [0.35], [-0.2]], device=DEV)      # rho(1), rho(1/2), rho(0): any three numbers
    yield g, x, _stack(sd, F, 3, 64, 1, True), sd, lut
    torch.cuda.empty_cache()


def test_full_size_sampled_rows_against_the_oracle(c4):
    from gnan_amd.functional import feature_mlps
    from gnan_amd.aggregate import rho_aggregate
    g, x, st, sd, lut = c4
    with torch.no_grad():
        S, total = feature_mlps(x, st, False, return_total=True)             # [N, 64]
        out = rho_aggregate(g, S, lut, True, s_total=total, reduce_channels=1)   # [N, 1]
        # the modules' default order (GNAN.py:157-170: feature sum first, then a ONE-column aggregation — the propagation-
        # blocked kernels of csrc/spmm_pb.hip at this size) against the same float64 rows
        s1, t1 = feature_mlps(x, st, True, return_total=True)                # [N, 1]
        out_sf = rho_aggregate(g, s1, lut, True, s_total=t1)
        del s1
    assert out.shape == (N, 1) and bool(torch.isfinite(out).all())
    assert out_sf.shape == (N, 1) and bool(torch.isfinite(out_sf).all())
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    rng = np.random.default_rng(0)
    mid = torch.nonzero((deg > 512) & (deg < 3000)).flatten()[:6].cpu().numpy()      # just over the hub-row threshold
    rows = np.unique(np.concatenate([rng.integers(0, N, 240), torch.topk(deg, 2).indices.cpu().numpy(), mid]))
    # shape functions at the sampled rows' neighbours: oracle restatement of GNAN.py:57-62 in float64
    scale = float(out.abs().max())
    worst = worst_sf = 0.0
    tot64 = total.double().cpu()
    cnt = g.cnt.cpu().numpy()
    p64 = {k: v.double() for k, v in sd.items()}
    for i in rows:
        lo, hi = int(g.rowptr[i]), int(g.rowptr[i + 1])
        cols = g.col[lo:hi].long()
        codes = g.code[lo:hi].long().cpu()
        fx = O.feature_mlps(x[cols].double().cpu(), p64).reshape(hi - lo, F)        # [deg, 64] float64
        w = lut.double().cpu().reshape(-1) / torch.from_numpy(np.maximum(cnt[i], 1)).double()
        acc = (w[codes].unsqueeze(1) * fx).sum(0) + w[-1] * (tot64 - fx.sum(0))
        worst = max(worst, abs(float(acc.sum()) - float(out[i, 0])))
        worst_sf = max(worst_sf, abs(float(acc.sum()) - float(out_sf[i, 0])))
    assert worst <= 1e-5 * scale, (worst, scale)
    assert worst_sf <= 1e-5 * scale, (worst_sf, scale)


def test_full_size_reference_order_equals_sum_first(c4):
    from gnan_amd.functional import feature_mlps
    from gnan_amd.aggregate import rho_aggregate
    g, x, st, sd, lut = c4
    with torch.no_grad():
        S, total = feature_mlps(x, st, False, return_total=True)
        ref_order = rho_aggregate(g, S, lut, True, s_total=total, reduce_channels=1)
        del S
        s1, t1 = feature_mlps(x, st, True, return_total=True)                # [N, 1]
        sum_first = rho_aggregate(g, s1, lut, True, s_total=t1)
    assert O.rel_err(ref_order.cpu(), sum_first.double().cpu()) <= 1e-5


def test_full_size_adjoint_linearity_and_row_subsets(c4):
    from gnan_amd.aggregate import spmm_launch
    g, x, st, sd, lut = c4
    gen = torch.Generator(device=DEV).manual_seed(3)
    W = 8
    s = torch.randn(N, W, generator=gen, device=DEV)
    s2 = torch.randn(N, W, generator=gen, device=DEV)
    y = torch.randn(N, W, generator=gen, device=DEV)
    As = spmm_launch(g, s, lut, True, False)                                  # listed pairs only: A is the weighted CSR
    # transposed use: weights are looked up by the neighbour (= the forward's row), as the dS path of the backward does
    Aty = spmm_launch(g.transposed(), y, lut, True, False, weight_by_col=True)
    lhs, rhs = float((y.double() * As.double()).sum()), float((Aty.double() * s.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), float((y.double().abs() * As.double().abs()).sum()))
    full = spmm_launch(g, 1.5 * s + s2, lut, True, True)
    parts = 1.5 * spmm_launch(g, s, lut, True, True) + spmm_launch(g, s2, lut, True, True)
    # three kernel results, each within the floor of its truth: |A(1.5 s + s2) - (1.5 A s + A s2)| <= 1e-5 (|.| + 1.5 |A s| + |A s2|)
    a1, a2 = spmm_launch(g, s, lut, True, True), spmm_launch(g, s2, lut, True, True)
    assert float((full - parts).abs().max()) <= 1e-5 * float(full.abs().max() + 1.5 * a1.abs().max() + a2.abs().max())
    ids = torch.randint(0, N, (50_000,), generator=gen, device=DEV).to(torch.int32)
    sub = spmm_launch(g, s, lut, True, True, row_ids=ids)
    ref = spmm_launch(g, s, lut, True, True)
    assert torch.equal(sub, ref[ids.long()])


def test_full_size_training_step_gradients(c4, monkeypatch):
    """Backward at the full size (sum-first, as the modules train): (i) the parameter gradients of the table path from
    gnan_fpwl_param_grads == the torch route's (probe points, float64); (ii) a directional derivative of the loss along a
    random parameter direction, by central differences of two more forwards, matches <grad, direction>."""
    from gnan_amd import functional
    from gnan_amd.functional import feature_mlps
    from gnan_amd.aggregate import rho_aggregate
    g, x, st, sd, lut = c4
    target = torch.randn(N, 1, generator=torch.Generator(device=DEV).manual_seed(9), device=DEV)

    def loss_of(stacked, lut_):
        S, total = feature_mlps(x, stacked, True, return_total=True)              # [N, 1]
        Y = rho_aggregate(g, S, lut_, True, s_total=total)
        return ((Y - target) ** 2).mean()

    grads = {}
    for tag, on in (("hip", True), ("torch", False)):
        monkeypatch.setattr(functional, "HIP_TABLE_GRADS", on)
        leaves = [t.detach().clone().requires_grad_(True) for t in st[:6]]
        lut_leaf = lut.detach().clone().requires_grad_(True)
        loss = loss_of(type(st)(*leaves, *st[6:]), lut_leaf)
        grads[tag] = torch.autograd.grad(loss, leaves + [lut_leaf])
    scale = max(float(t.abs().max()) for t in grads["torch"][:6])
    for a, b in zip(grads["hip"], grads["torch"]):
        assert float((a - b).abs().max()) <= 1e-5 * scale
    monkeypatch.setattr(functional, "HIP_TABLE_GRADS", True)
    gen = torch.Generator(device=DEV).manual_seed(4)
    direction = [torch.randn(t.shape, generator=gen, device=DEV) for t in st[:6]] + [torch.randn(lut.shape, generator=gen, device=DEV)]
    analytic = sum(float((a.double() * d.double()).sum()) for a, d in zip(grads["hip"], direction))
    eps = 1e-3
    with torch.no_grad():
        up = loss_of(type(st)(*[t + eps * d for t, d in zip(st[:6], direction)], *st[6:]), lut + eps * direction[6])
        dn = loss_of(type(st)(*[t - eps * d for t, d in zip(st[:6], direction)], *st[6:]), lut - eps * direction[6])
    numeric = (float(up) - float(dn)) / (2 * eps)
    assert abs(numeric - analytic) <= 2e-2 * max(abs(analytic), 1e-6), (numeric, analytic)


def test_full_size_pre_rho_normalisation(c4, monkeypatch):
    """The stand-alone TensorGNAN (GNAN.py:55-79: rho(node_distances / normalization_matrix)) at the full size: the per-row
    weight table comes from gnan_rho_row_lut (rho's exact table, 3 look-ups per row); sampled rows against a float64
    restatement; and the forward costs what the post-rho forward costs (the table replaces the count table one for one)."""
    import time
    from gnan_amd import GNAN as standalone
    from gnan_amd import models, replay
    monkeypatch.setattr(replay, "REPLAY_FORWARD", False)     # (the launches' cost is what is compared: a replayed forward skips their host side)
    g, x, st, sd, lut = c4
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(1)
    mods = {}
    for name, cls in (("pre", standalone.TensorGNAN), ("post", models.TensorGNAN)):
        m = cls(F, 1, 3, hidden_channels=64, device=DEV)
        with torch.no_grad():
            for _, p in m.named_parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else (2.0 / sum(p.shape)) ** 0.5))
        mods[name] = m.to(DEV).eval()

    class Bag:
        pass
    data = Bag()
    data.x, data.edge_index, data.gnan_graph = x, None, g
    times = {}
    with torch.no_grad():
        for name, m in mods.items():
            for _ in range(3):
                out = m.forward(data)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                out = m.forward(data)
            torch.cuda.synchronize()
            times[name] = (time.perf_counter() - t0) / 10
            if name == "pre":
                y = out
    # The assertion guards the route, not the last per cent (a torch MLP on N x D rows would be tens of milliseconds).  Since round 6
    # the post-rho forward takes the propagation-blocked aggregation (1.0 ms) and the per-row table of the pre-rho class keeps the
    # row-parallel kernels (1.8-1.9 ms): generous enough for that and for a noisy shared box
    assert times["pre"] <= 2.5 * times["post"] + 5e-4, times
    m = mods["pre"]
    p64 = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(0)
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    rows = np.unique(np.concatenate([rng.integers(0, N, 120), torch.topk(deg, 2).indices.cpu().numpy()]))
    with torch.no_grad():
        S, total = m._operand(x, "fs", m.fs, True, True)                 # [N, 1] and its column sum, as the forward forms them
    tot64 = total.double().cpu()
    cnt = g.cnt.cpu().numpy()
    worst = 0.0
    for i in rows:
        lo, hi = int(g.rowptr[i]), int(g.rowptr[i + 1])
        cols = g.col[lo:hi].long()
        codes = g.code[lo:hi].long().cpu()
        fx = O.feature_mlps(x[cols].double().cpu(), p64).sum(1)                                  # [deg, 1]
        w = O.row_lut_pre_rho(p64, cnt[i:i + 1], torch.float64)[0]                               # [3, 1]
        acc = (w[codes] * fx).sum(0) + w[-1] * (tot64 - fx.sum(0))
        worst = max(worst, float((acc - y[i].double().cpu()).abs().max()))
    assert worst <= 1e-5 * float(y.abs().max()), (worst, float(y.abs().max()))
