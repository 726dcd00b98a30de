"""BASELINE.json's configurations that no other GPU test runs at their own shape.

* C1 — Cora-shaped dense inputs (N = 2708, F = 1433 + 1, C = 7, H = 64, L = 3; bag-of-words rows, row-normalised, ones
  column) through ``models.GNAN`` (what main.py:79 picks) and ``models.TensorGNAN``: forward on a sample of nodes and one
  backward against the float64 oracle's per-node loop (GNAN.py:146-172).  TensorGNAN keeps the reference's zero biases
  (GNAN.py:49-53), so most inputs sit exactly on the kinks of their shape functions.
* C3 — ogbn-arxiv-shaped (datasets.py:273-291: N = 169 343, E = 1 166 243, F = 128 + 1; C = 1 as the reference sets it and
  C = 40, the data set's class count): preferential-attachment edges + self pairs, K = 1, ``models.TensorGNAN`` forward +
  backward at the configuration's OWN shape — the WHOLE output (the 8 193-citation hub included) and EVERY parameter
  gradient against float64 oracle autograd (shape functions back-propagated in node chunks).
* C5 — papers100M-shaped R-MAT (scale 27, 111M nodes / 1.6G edges), bf16 operand storage: sampled rows against a
  float64 restatement on the rounded operand, reference order == sum-first, row subsets bit-identical.
* more than 2^31 listed pairs (int64 row offsets, pair indices beyond 32 bits, a hub row at the far end): sampled rows
  against the float64 restatement, degree-sorted / packed-index / hot-column walks == the natural-order walk.
"""
import numpy as np
import pytest
import torch

from helpers import grad_rule, module_grads, oracle_grads
from oracle import gnan_oracle as O
from test_gpu_kernels import _mlp_state, _stack

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# ---------------------------------------------------------------------------------------------------------- C1
@pytest.fixture(scope="module")
def cora_shaped():
    rng = np.random.default_rng(0)
    n, f_raw = 2708, 1433
    ei = rng.integers(0, n, (2, 5278))
    ei = ei[:, ei[0] != ei[1]]
    ei = np.unique(np.concatenate([ei, ei[::-1]], axis=1), axis=1)
    nd, norm = O.pre_process_dense(ei, n)
    words = (rng.random((n, f_raw)) < 0.0127).astype(np.float32)           # ~18 words per document, like Cora
    words[words.sum(1) == 0, 0] = 1.0
    x = words / words.sum(1, keepdims=True)                                 # T.NormalizeFeatures (datasets.py:94)
    x = torch.from_numpy(np.concatenate([x, np.ones((n, 1), np.float32)], axis=1))   # pre_process_datasets.py:127
    return x, nd, norm


@pytest.mark.parametrize("cls", ["GNAN", "TensorGNAN", "TensorGNAN_reference_order"])
def test_c1_cora_shaped_dense_inputs(cora_shaped, cls):
    from gnan_amd import models
    x, nd, norm = cora_shaped
    n, F = x.shape
    C, H, L = 7, 64, 3
    torch.manual_seed(1)
    if cls == "GNAN":
        mod = models.GNAN(F, C, num_layers=L, hidden_channels=H, device=DEV)               # default init, non-zero biases
    else:
        mod = models.TensorGNAN(F, C, L, hidden_channels=H, device=DEV)
        with torch.no_grad():                              # O(1) weights; biases stay 0 as the reference leaves them
            for name, p in mod.named_parameters():
                if p.dim() == 2:
                    torch.nn.init.xavier_normal_(p, gain=1.0)
        if cls.endswith("reference_order"):
            mod.aggregation_order = "reference"
    sd64 = {k: v.detach().double().clone() for k, v in mod.state_dict().items()}
    mod = mod.to(DEV).eval()
    data = Bag(x=x.to(DEV), edge_index=None, node_distances=nd.to(DEV), normalization_matrix=norm.to(DEV))
    ids = np.random.default_rng(5).choice(n, 48, replace=False)
    target = torch.randn(len(ids), C, generator=torch.Generator().manual_seed(2), dtype=torch.float64)

    y = mod.forward(data)
    assert y.shape == (n, C)
    loss = ((y[torch.from_numpy(ids).to(DEV)] - target.to(DEV).float()) ** 2).sum()
    loss.backward()

    with torch.no_grad():
        truth = O.gnan_forward(x.double(), nd.double(), norm.double(), sd64, True, node_ids=ids.tolist())   # [48, C]
    assert O.rel_err(y.detach().cpu()[ids], truth) <= 1e-5
    g64 = oracle_grads(lambda p: ((O.gnan_forward(x.double(), nd.double(), norm.double(), p, True, node_ids=ids.tolist())
                                   - target) ** 2).sum(), sd64, torch.float64)
    g32 = lambda: oracle_grads(lambda p: ((O.gnan_forward(x, nd, norm, p, True, node_ids=ids.tolist()) - target.float()) ** 2).sum(),
                               sd64, torch.float32)                               # (evaluated only if the floor does not decide)
    ok, e_build, e_ref, where = grad_rule(module_grads(mod), g64, g32)              # SURVEY 8c on the gradient: max(1e-5, fp32 oracle's own)
    assert ok, f"{where}: build {e_build:.3e} vs fp32 oracle {e_ref:.3e}"

    if cls == "GNAN":                                       # the per-node signature (GNAN.py:146): rows on request
        with torch.no_grad():
            sub = mod.forward(data, ids.tolist())
        # (dense rows are sliced over workgroups by the number of requested rows: same sums, another association)
        assert O.rel_err(sub.cpu(), y.detach().cpu()[ids].double()) <= 2e-6
        assert O.rel_err(sub.cpu(), truth.detach()) <= 1e-5


# ---------------------------------------------------------------------------------------------------------- C3
@pytest.mark.parametrize("C", [1, 40])
def test_c3_arxiv_shaped_forward_backward_at_its_own_shape(C):
    import gnan_amd  # noqa: F401
    from gnan_amd import models
    from gnan_amd import synthetic as syn
    N, E, F, H, L = 169_343, 1_166_243, 129, 64, 3
    src, dst = syn.preferential_attachment_edges(N, E, seed=0, device=DEV)
    indeg = torch.bincount(dst, minlength=N)
    assert int(indeg.max()) > 2000                          # a hub: its operand row is gathered by thousands of rows
    g = syn.hop1_csr(src, dst, N)
    assert g.nnz == E + N and g.n_codes == 3
    x = syn.block_features(N, F, 0, N, seed=1, device=DEV)
    torch.manual_seed(0)
    mod = models.TensorGNAN(F, C, L, hidden_channels=H, device=DEV)
    gen = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else (2.0 / sum(p.shape)) ** 0.5))
    sd = {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}
    mod = mod.to(DEV).eval()
    target = torch.randn(N, C, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    data = Bag(x=x, edge_index=None, gnan_graph=g)

    y = mod.forward(data)
    assert y.shape == (N, C)
    ((y - target.to(DEV).float()) ** 2).mean().backward()

    # ---- float64 truth: oracle shape functions (GNAN.py:57-62) in node chunks, shell-form aggregation
    rowptr, col, code = g.rowptr.cpu().long().numpy(), g.col.cpu().numpy(), g.code.cpu().numpy()
    cnt = g.cnt.cpu().long().numpy()
    xh = x.cpu()
    p64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    chunk = 16384
    with torch.no_grad():
        S64 = torch.cat([O.feature_mlps(xh[i:i + chunk].double(), p64).sum(1) for i in range(0, N, chunk)])     # [N, C]
        S32 = torch.cat([O.feature_mlps(xh[i:i + chunk], sd).sum(1) for i in range(0, N, chunk)])
        ref32 = O.spmm_csr_vectorised(rowptr, col, code, S32, O.rho_lut(sd, 3), cnt)
    S_leaf = S64.clone().requires_grad_(True)
    truth = O.spmm_csr_vectorised(rowptr, col, code, S_leaf, O.rho_lut(p64, 3, torch.float64), cnt)
    e_ref = O.rel_err(ref32, truth.detach())                # the tolerance rule: max(1e-5, the fp32 reference's error)
    err = O.rel_err(y.detach().cpu(), truth.detach())
    assert err <= max(1e-5, e_ref), (err, e_ref)
    hub = int(torch.argmax(indeg))
    rows_of_hub = torch.nonzero(dst == hub).flatten()[:64]
    sub = src[rows_of_hub].cpu()
    assert O.rel_err(y.detach().cpu()[sub], truth.detach()[sub]) <= max(1e-5, e_ref)

    ((truth - target) ** 2).mean().backward()               # rho's parameters and dS
    dS = S_leaf.grad
    for i in range(0, N, chunk):                            # the shape functions' parameters, chunk by chunk
        O.feature_mlps(xh[i:i + chunk].double(), p64).sum(1).backward(dS[i:i + chunk])
    def reference32():
        """The float32 reference of the same chain (its own arithmetic restated by the oracle, chunked the same way) — only
        evaluated when the build's gradient is not within the floor of the float64 truth anyway."""
        p32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        S32_leaf = S32.clone().requires_grad_(True)
        out32 = O.spmm_csr_vectorised(rowptr, col, code, S32_leaf, O.rho_lut(p32, 3), cnt)
        ((out32 - target.float()) ** 2).mean().backward()
        for i in range(0, N, chunk):
            O.feature_mlps(xh[i:i + chunk], p32).sum(1).backward(S32_leaf.grad[i:i + chunk])
        return {k: v.grad for k, v in p32.items()}
    ok, e_build, e_g32, where = grad_rule(module_grads(mod), {k: v.grad for k, v in p64.items()}, reference32)
    assert ok, f"{where}: build {e_build:.3e} vs fp32 oracle {e_g32:.3e}"                    # SURVEY 8c on the gradient


# ---------------------------------------------------------------------------------------------------------- sampled rows
def _sampled_rows_check(g, operand, total, lut, out, rows, reduce):
    """``out[i] = sum_e w(i, code_e) S[col_e] + w(i, rest) (total - sum_e S[col_e])`` in float64 on the host, operand rows
    fetched from the device (bf16 rows as stored: the oracle runs on the rounded operand)."""
    cnt = g.cnt
    lut64 = lut.double().cpu().reshape(-1)
    tot64 = total.double().cpu()
    worst = 0.0
    for i in rows:
        lo, hi = int(g.rowptr[i]), int(g.rowptr[i + 1])
        cols = g.col[lo:hi].long()
        codes = g.code[lo:hi].long().cpu()
        rowsS = operand[cols].double().cpu()                                           # [deg, W]
        w = lut64 / torch.clamp(cnt[i].double().cpu(), min=1.0)
        acc = (w[codes].unsqueeze(1) * rowsS).sum(0) + w[-1] * (tot64 - rowsS.sum(0))
        want = float(acc.sum()) if reduce else acc
        got = float(out[i, 0]) if reduce else out[i].double().cpu()
        worst = max(worst, abs(want - got) if reduce else float((want - got).abs().max()))
    return worst


# ---------------------------------------------------------------------------------------------------------- C5
@pytest.fixture(scope="module")
def c5():
    import gnan_amd  # noqa: F401
    from gnan_amd import synthetic as syn
    N, E, F = 111_059_956, 1_615_685_872, 64
    src, dst = syn.rmat_edges(27, N, E, seed=0, device=DEV)
    g = syn.hop1_csr(src, dst, N)
    del src, dst
    torch.cuda.empty_cache()
    x = syn.block_features(N, F, 0, N, seed=1, device=DEV)
    sd = _mlp_state(F, 3, 64, 1, True, seed=5)
    lut = torch.tensor([[0.9], [0.35], [-0.2]], device=DEV)
    yield N, g, x, _stack(sd, F, 3, 64, 1, True), lut
    del g, x
    torch.cuda.empty_cache()


def test_c5_papers100m_shaped_bf16(c5):
    from gnan_amd.functional import feature_mlps
    from gnan_amd.aggregate import rho_aggregate, spmm_launch
    N, g, x, st, lut = c5
    assert g.nnz == 111_059_956 + 1_615_685_872 and g.nnz * 4 > 2 ** 32      # byte offsets of the index arrays exceed 32 bits
    with torch.no_grad():
        S, total = feature_mlps(x, st, False, return_total=True, out_dtype=torch.bfloat16)        # [N, 64] bf16
        assert S.dtype == torch.bfloat16
        out = rho_aggregate(g, S, lut, True, s_total=total, reduce_channels=1)                     # [N, 1]
    assert out.shape == (N, 1) and bool(torch.isfinite(out).all())
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    rng = np.random.default_rng(0)
    rows = np.unique(np.concatenate([rng.integers(0, N, 90), torch.topk(deg, 2).indices.cpu().numpy(),
                                     np.arange(N - 4, N), np.arange(4)]))
    scale = float(out.abs().max())
    assert _sampled_rows_check(g, S, total, lut, out, rows, True) <= 1e-5 * scale
    with torch.no_grad():
        # reference order on the bf16 operand vs sum-first in fp32: the storage format's own rounding separates them, not
        # the kernels (the check above is the kernels' — 1e-5 against float64 on the operand AS STORED).  A stored value is
        # off by up to 2^-9 of itself; a row adds 64 x (1 + degree) of them with weights <= 0.9: ~0.02 absolute at degree 1,
        # against outputs of 5..9, and the maximum is taken over 111M rows (measured 5.4e-3 of the largest output)
        s1, t1 = feature_mlps(x, st, True, return_total=True)
        sum_first = rho_aggregate(g, s1, lut, True, s_total=t1)
        assert O.rel_err(out.cpu(), sum_first.double().cpu()) <= 1e-2
        ids = torch.from_numpy(rows.astype(np.int32)).to(DEV)
        sub = spmm_launch(g, S, lut, True, True, row_ids=ids, s_total=total, reduce_cr=1)
        assert torch.equal(sub, out[ids.long()])                                                   # row subsets: bit-identical
        del S
        worst = _sampled_rows_check(g, s1, t1, lut, sum_first, rows[:40], False)
        assert worst <= 1e-5 * float(sum_first.abs().max())


# ---------------------------------------------------------------------------------------------------------- > 2^31 pairs
def test_more_than_2_31_listed_pairs(monkeypatch):
    """int64 row offsets and pair indices beyond 32 bits: 30M rows of 50..100 pairs (2.25G pairs) and a 3M-pair hub row
    whose slices start beyond pair 2^31."""
    import gnan_amd  # noqa: F401
    from gnan_amd import HopGraph, functional
    from gnan_amd.functional import column_sums
    from gnan_amd.aggregate import spmm_launch
    N = 30_000_000
    ar = torch.arange(N, device=DEV)
    deg = 50 + (ar % 51)
    hub = N - 5
    deg[hub] = 3_000_000
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=DEV)
    rowptr[1:] = torch.cumsum(deg, 0)
    nnz = int(rowptr[-1])
    assert nnz > 2 ** 31 + 2 ** 26
    gen = torch.Generator(device=DEV).manual_seed(0)
    col = torch.empty(nnz, dtype=torch.int32, device=DEV)
    for e0 in range(0, nnz, 1 << 28):
        e1 = min(nnz, e0 + (1 << 28))
        col[e0:e1] = torch.randint(0, N, (e1 - e0,), generator=gen, device=DEV, dtype=torch.int32)
    code = torch.ones(nnz, dtype=torch.uint8, device=DEV)
    col[rowptr[:-1]] = ar.to(torch.int32)                                     # the self pair leads every row
    code[rowptr[:-1]] = 0
    cnt = torch.stack([torch.ones_like(deg), deg - 1, N - deg], dim=1).to(torch.int32)
    g = HopGraph.from_csr(rowptr, col, code, n_cols=N, n_codes=3, cnt=cnt)
    assert g.rowptr.dtype == torch.int64
    lut = torch.tensor([[0.9], [0.35], [-0.2]], device=DEV)
    rng = np.random.default_rng(1)
    rows = np.unique(np.concatenate([rng.integers(0, N, 40), np.arange(N - 8, N), np.arange(3), [hub]]))
    assert int(rowptr[rows[-1]]) > 2 ** 31
    ids = torch.from_numpy(rows.astype(np.int32)).to(DEV)
    for W, dtype in ((64, torch.bfloat16), (64, torch.float32), (1, torch.float32), (2, torch.float32)):
        S = torch.empty((N, W), dtype=dtype, device=DEV)
        for r0 in range(0, N, 1 << 22):
            S[r0:r0 + (1 << 22)] = torch.rand((min(N, r0 + (1 << 22)) - r0, W), generator=gen, device=DEV).to(dtype)
        total = column_sums(S)
        reduce = 1 if W == 64 else 0
        out = spmm_launch(g, S, lut, True, True, s_total=total, reduce_cr=reduce)
        scale = float(out.abs().max())
        assert _sampled_rows_check(g, S, total, lut, out, rows, bool(reduce)) <= 1e-5 * scale, (W, dtype)
        sub = spmm_launch(g, S, lut, True, True, row_ids=ids, s_total=total, reduce_cr=reduce)     # natural order, by row id
        assert torch.equal(sub, out[ids.long()]), (W, dtype)
        del S, out, sub
    del g, col, code
    torch.cuda.empty_cache()


def test_c3_passes_that_finish_their_own_sums_equal_the_two_launch_routes(monkeypatch):
    """Round 6: with an arrival counter the last workgroup of a pass takes the sum over the workgroups' partial results (moment
    scales, the group-sum's column total, the packed rows' q), and the look-up's bucket tables come out of the table build's
    compaction pass.  Same bits as the extra launches they replace — outputs and every gradient, eagerly and replayed from the
    captured step (counters in the step's zeroed scratch)."""
    import gnan_amd  # noqa: F401
    from gnan_amd import functional as Fn, models, pwl
    from gnan_amd import synthetic as syn
    N, E, F, H, L = 169_343, 1_166_243, 129, 64, 3
    src, dst = syn.preferential_attachment_edges(N, E, seed=0, device=DEV)
    g = syn.hop1_csr(src, dst, N)
    x = syn.block_features(N, F, 0, N, seed=1, device=DEV)
    target = torch.randn(N, 1, generator=torch.Generator().manual_seed(2)).to(DEV)
    data = Bag(x=x, edge_index=None, gnan_graph=g)

    def run(fused, steps):
        monkeypatch.setattr(Fn, "ARRIVE_COUNTERS", fused)
        monkeypatch.setattr(pwl, "INDEX_IN_BUILD", fused)
        torch.manual_seed(0)
        mod = models.TensorGNAN(F, 1, L, hidden_channels=H, device=DEV)
        gen = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for _, p in mod.named_parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else (2.0 / sum(p.shape)) ** 0.5))
        mod = mod.to(DEV).train()
        outs, launches = [], []
        for _ in range(steps):                                  # the third call on the same inputs is a replay (gnan_amd/replay.py)
            mod.zero_grad(set_to_none=True)
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                y = mod.forward(data)
                ((y - target) ** 2).mean().backward()
                torch.cuda.synchronize()
            launches.append(sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in e.name
                                and "Memset" not in e.name))
            outs.append((y.detach().clone(), {k: v.clone() for k, v in module_grads(mod).items()}))
        return outs, launches

    fused, n_fused = run(True, 4)
    plain, n_plain = run(False, 4)
    for (ya, ga), (yb, gb) in zip(fused, plain):
        assert torch.equal(ya, yb)
        assert ga.keys() == gb.keys() and all(torch.equal(ga[k], gb[k]) for k in ga), [k for k in ga if not torch.equal(ga[k], gb[k])]
    # an eager step; a replayed one.  (At this size only the scales and the bucket tables lose their launch: a pass of 662
    # workgroups keeps its second launch — an agent-scope fence per workgroup cost 13 us against the 5 it saves; the group-sum
    # total and the packed rows' q use their counters below 128 workgroups, e.g. on Cora-sized graphs.)
    assert n_plain[1] - n_fused[1] >= 2 and n_plain[3] - n_fused[3] >= 2, (n_plain, n_fused)
