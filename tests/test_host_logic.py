"""CPU tests of host-side logic that needs no kernel: the deferred rest-bucket term, the degree-sorted CSR copy, the
padding of ragged feature counts, the speculative table sizes (gnan_amd/functional.py, graph.py, pwl.py)."""
import numpy as np
import pytest
import torch

import gnan_amd  # noqa: F401
from gnan_amd import HopGraph, aggregate, functional, pwl
from gnan_amd.functional import StackedMLP
from oracle import gnan_oracle as O


def _csr(n_rows, n_cols, K, rng, hubs=()):
    deg = rng.poisson(5, n_rows)
    deg[rng.random(n_rows) < 0.15] = 0
    for r, d in hubs:
        deg[r] = d
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n_cols, int(rowptr[-1])).astype(np.int32)
    code = rng.integers(0, K + 1, int(rowptr[-1])).astype(np.uint8)
    return rowptr, col, code


@pytest.mark.parametrize("W,Cw,cr,per_row,use_cnt", [(8, 1, 1, False, True), (8, 1, 0, False, True), (12, 4, 4, False, True),
                                                      (12, 4, 2, False, False), (6, 3, 0, True, False), (1, 1, 0, False, True)])
def test_rest_total_term_is_the_rest_bucket_weight_times_the_column_sums(W, Cw, cr, per_row, use_cnt):
    rng = np.random.default_rng(W + Cw + cr)
    n, K = 50, 1
    D = K + 2
    rowptr, col, code = _csr(n, n, K, rng)
    g = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(code), n_cols=n, n_codes=D)
    lut = torch.from_numpy(rng.standard_normal((n, D, Cw) if per_row else (D, Cw)).astype(np.float32))
    total = torch.from_numpy(rng.standard_normal(W).astype(np.float32))
    import cpu_kernels                                     # the stand-in the gloo tests use (the kernel: tests/test_gpu_kernels.py)
    got = cpu_kernels.rest_total_term(g, lut, use_cnt, total, cr)
    w_rest = (lut[:, D - 1, :] if per_row else lut[D - 1].expand(n, Cw)).double()
    if use_cnt:
        w_rest = w_rest / g.cnt[:, D - 1:D].clamp_min(1).double()
    full = w_rest[:, torch.arange(W) % Cw] * total.double()                 # [n, W]
    want = full.view(n, W // cr, cr).sum(1) if cr else full
    assert got.shape == want.shape
    assert O.rel_err(got, want) <= 1e-6


def test_degree_sorted_copy_is_a_row_permutation_of_the_csr():
    rng = np.random.default_rng(3)
    n, K = 400, 2
    rowptr, col, code = _csr(n, n, K, rng, hubs=[(7, 900), (300, 40)])
    g = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(code), n_cols=n, n_codes=K + 2)
    gs, order, plan = g.degree_sorted_copy()
    o = order.numpy()
    deg = np.diff(rowptr)
    assert np.array_equal(np.sort(o), np.arange(n)) and np.all(np.diff(deg[o]) >= 0)
    rp = gs.rowptr.numpy()
    assert np.array_equal(np.diff(rp), deg[o]) and gs.rowptr.dtype == g.rowptr.dtype
    for q in range(n):
        a, b = rowptr[o[q]], rowptr[o[q] + 1]
        assert np.array_equal(gs.col.numpy()[rp[q]:rp[q + 1]], col[a:b])
        assert np.array_equal(gs.code.numpy()[rp[q]:rp[q + 1]], code[a:b])
    assert torch.equal(gs.cnt, g.cnt[order.long()])
    assert plan.n_long == 1 and int(plan.rows[0]) == n - 1                    # the 900-pair row is last in degree order
    assert g.degree_sorted_copy()[0] is gs                                    # cached


def test_padded_stack_appends_zero_functions():
    F, H, C, Fp = 5, 4, 1, 16
    g = torch.Generator().manual_seed(0)
    p = StackedMLP(torch.randn(F, H, generator=g), torch.randn(F, H, generator=g), torch.randn(1, F, H, H, generator=g),
                   None, torch.randn(F, C, H, generator=g), torch.randn(F, C, generator=g), 3, H, C, F)
    pp = functional._padded_stack(p, Fp)
    assert pp.F == Fp and pp.w_first.shape == (Fp, H) and pp.w_mid.shape == (1, Fp, H, H) and pp.b_mid is None
    assert pp.w_last.shape == (Fp, C, H) and pp.b_last.shape == (Fp, C)
    assert torch.equal(pp.w_first[:F], p.w_first) and float(pp.w_first[F:].abs().max()) == 0.0
    assert torch.equal(pp.w_mid[:, :F], p.w_mid) and float(pp.w_mid[:, F:].abs().max()) == 0.0
    assert float(pp.w_last[F:].abs().max()) == 0.0 and float(pp.b_last[F:].abs().max()) == 0.0
    x = torch.rand(7, F)
    xp = functional._padded_x(x, Fp)
    assert xp.shape == (7, Fp) and torch.equal(xp[:, :F], x) and float(xp[:, F:].abs().max()) == 0.0
    assert functional._padded_x(x, Fp) is xp                                  # cached per (static) feature matrix


def test_speculative_table_sizes_cover_only_what_they_can_hold():
    exact = pwl.PwlTables(None, None, None, None, 130, 16, 2080)
    assert pwl.covers(pwl.PwlTables(None, None, None, None, 256, 16, 2348), exact)
    assert not pwl.covers(pwl.PwlTables(None, None, None, None, 128, 16, 2348), exact)      # search too shallow
    assert not pwl.covers(pwl.PwlTables(None, None, None, None, 256, 16, 2000), exact)      # LDS image too small
    assert pwl.covers(pwl.PwlTables(None, None, None, None, 256, 8, 8 * 130), exact)        # another grouping, worst case fits
    assert not pwl.covers(pwl.PwlTables(None, None, None, None, 256, 8, 8 * 130 - 1), exact)


def test_tensor_keyed_cache_matches_objects_not_addresses():
    """The reference loops upload one graph per step (trainer.py:46); the allocator recycles the freed graph's addresses
    for the next one of the same size, so a derived value may only be found again through the tensor OBJECT it came from."""
    import copy
    import gc
    import pickle
    from gnan_amd._cache import TensorKeyedCache
    c = TensorKeyedCache(3)
    a = torch.zeros(4, 4)
    assert c.get((a, None), "dense") is None
    assert c.put((a, None), "dense", "graph of a") == "graph of a"
    assert c.get((a, None), "dense") == "graph of a"
    assert c.get((a, None), "csr") is None and c.get((a,), "dense") is None and c.get((a, a), "dense") is None
    twin = a.detach()                                  # same storage, same data_ptr, same version counter: another object
    assert twin.data_ptr() == a.data_ptr() and c.get((twin, None), "dense") is None
    a.add_(1)                                          # in-place write: the derived value is stale
    assert c.get((a, None), "dense") is None and len(c) == 0
    c.put((a, None), "dense", "graph of a'")
    twin.mul_(2)                                       # ... also through an alias (views share the version counter)
    assert c.get((a, None), "dense") is None
    c.put((a, None), "dense", 1)
    del a, twin
    gc.collect()
    assert len(c) == 0                                 # the entry went with its source
    b = torch.zeros(4, 4)                              # may or may not land on the old address / id: must miss either way
    assert c.get((b, None), "dense") is None
    keep = [torch.zeros(2) for _ in range(5)]
    for i, t in enumerate(keep):
        c.put((t,), None, i)
    assert len(c) == 3 and c.get((keep[0],)) is None and c.get((keep[4],)) == 4      # least recently used dropped
    assert len(copy.deepcopy(c)) == 0 and len(pickle.loads(pickle.dumps(c))) == 0


def test_padded_feature_matrix_is_not_reused_for_another_tensor():
    Fp = 16
    x1 = torch.rand(6, 5)
    p1 = functional._padded_x(x1, Fp)
    x2 = x1.detach()                                   # the look-alike a recycled allocation would be: same address, version, shape
    x2_expected = torch.nn.functional.pad(x2, (0, Fp - 5))
    assert functional._padded_x(x2, Fp) is not p1 and torch.equal(functional._padded_x(x2, Fp), x2_expected)
    x1.mul_(3)
    assert torch.equal(functional._padded_x(x1, Fp)[:, :5], x1)


def test_degree_sorted_copy_of_a_transposed_rectangular_graph_keeps_the_count_table():
    """A transposed graph carries the forward graph's count table (rows = its neighbours); the degree-sorted copy the
    narrow-operand backward asks for (s_by_code) must leave it alone — permuting it with the rows indexed out of range
    whenever the graph is rectangular (halo or row-partitioned graphs: more operand rows than output rows)."""
    rng = np.random.default_rng(5)
    n_rows, n_cols, K = 60, 150, 1
    rowptr, col, code = _csr(n_rows, n_cols, K, rng)
    g = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(code), n_cols=n_cols, n_codes=K + 2)
    t = g.transposed()
    assert (t.n_rows, t.n_cols) == (n_cols, n_rows) and t.cnt is g.cnt
    ts, order, _ = t.degree_sorted_copy()
    assert ts.cnt is g.cnt and ts.n_rows == n_cols
    o = order.numpy()
    trp, tsrp = t.rowptr.numpy(), ts.rowptr.numpy()
    for q in range(n_cols):
        a, b = trp[o[q]], trp[o[q] + 1]
        assert np.array_equal(ts.col.numpy()[tsrp[q]:tsrp[q + 1]], t.col.numpy()[a:b])
    sq = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col % n_rows), torch.from_numpy(code), n_cols=n_rows,
                           n_codes=K + 2)
    assert sq.transposed().degree_sorted_copy()[0].cnt is sq.cnt               # square: same rule, no meaningless permutation


def test_rows_helper_materialises_expanded_and_transposed_views():
    g = torch.ones(3).expand(5, 3)                                             # strides (0, 1): what .sum(0) sends back
    assert g.stride() == (0, 1) and functional._rows(g).stride() == (3, 1)
    t = torch.rand(4, 6).t()
    assert functional._rows(t).stride() == (4, 1)
    v = torch.rand(8, 10)[:, :6]                                               # wide row stride: fine as it is
    assert functional._rows(v) is v
    one = torch.ones(3).expand(1, 3)
    assert functional._rows(one) is one


@pytest.fixture
def cpu_kernels():
    import cpu_kernels as ck
    undo = ck.install()
    yield ck
    undo()


@pytest.mark.parametrize("W,Cw,K,use_cnt,per_row,cr,subset", [
    (4, 1, 1, True, False, 0, False),        # narrow operand: pre-weighted (node, hop code) rows, fused table gradient
    (48, 1, 1, True, False, 1, False),       # wide operand with the fused feature sum
    (6, 3, 2, True, False, 0, False),        # several weight channels: shell-sums route
    (6, 3, 1, False, True, 0, False),        # per-row tables (pre-rho normalisation)
    (8, 1, 4, True, False, 0, True),         # more shells than the fused table gradient takes, row subset
    (2, 1, 1, True, False, 0, True),
])
def test_aggregation_backward_host_logic_vs_oracle_autograd(cpu_kernels, W, Cw, K, use_cnt, per_row, cr, subset):
    """aggregate._RhoAggregate on stand-in launchers (tests/cpu_kernels.py) == autograd through the oracle: checks which
    launches the backward pass makes and what it adds around them (rest-bucket terms, fused-sum broadcast, row subsets)."""
    rng = np.random.default_rng(W * 7 + Cw + K)
    n, D = 40, K + 2
    rowptr, col, code = _csr(n, n, K, rng, hubs=[(3, 30)])
    g = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(code), n_cols=n, n_codes=D)
    S = torch.from_numpy(rng.standard_normal((n, W)).astype(np.float32)).requires_grad_(True)
    lut = torch.from_numpy(rng.standard_normal((n, D, Cw) if per_row else (D, Cw)).astype(np.float32)).requires_grad_(True)
    rows = torch.from_numpy(rng.permutation(n)[:11].astype(np.int32)) if subset else None
    Y = aggregate.rho_aggregate(g, S, lut, use_cnt, row_ids=rows, reduce_channels=cr)
    up = torch.from_numpy(rng.standard_normal(tuple(Y.shape)).astype(np.float32))
    dS, dlut = torch.autograd.grad(Y, [S, lut], up)
    S64, lut64 = S.detach().double().requires_grad_(True), lut.detach().double().requires_grad_(True)
    wt = lut64 if per_row else lut64.unsqueeze(0).expand(n, -1, -1)
    if use_cnt:
        wt = wt / g.cnt.clamp_min(1).double().unsqueeze(-1)
    want = O.spmm_csr(rowptr, col, code, S64, wt)
    if rows is not None:
        want = want[rows.long()]
    if cr:
        want = want.view(want.shape[0], -1, cr).sum(1)
    assert O.rel_err(Y.detach(), want.detach()) <= 1e-5
    dS64, dlut64 = torch.autograd.grad(want, [S64, lut64], up.double())
    assert O.rel_err(dS, dS64) <= 1e-5 and O.rel_err(dlut, dlut64) <= 1e-5


def test_rest_bucket_total_over_the_first_rows_only(cpu_kernels):
    """``total_rows``: only the first rows of the operand were summed into ``s_total`` (owned rows ahead of halo rows), so
    only they receive the rest-bucket part of the operand gradient."""
    rng = np.random.default_rng(9)
    n_rows, n_cols, n_own, W, D = 30, 50, 30, 8, 3
    rowptr, col, code = _csr(n_rows, n_cols, 1, rng)
    g = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(code), n_cols=n_cols, n_codes=D)
    S = torch.from_numpy(rng.standard_normal((n_cols, W)).astype(np.float32)).requires_grad_(True)
    lut = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).requires_grad_(True)
    total = S[:n_own].sum(0).detach()
    Y = aggregate.rho_aggregate(g, S, lut, True, s_total=total, total_rows=n_own)
    up = torch.from_numpy(rng.standard_normal(tuple(Y.shape)).astype(np.float32))
    dS, dlut = torch.autograd.grad(Y, [S, lut], up)
    S64, lut64 = S.detach().double().requires_grad_(True), lut.detach().double().requires_grad_(True)
    wt = lut64.unsqueeze(0) / g.cnt.clamp_min(1).double().unsqueeze(-1)
    listed = O.spmm_csr(rowptr, col, code, S64, wt, with_rest=False)
    ones = torch.ones_like(wt)
    plain = O.spmm_csr(rowptr, col, code, S64, ones, with_rest=False)              # unweighted sum of the listed rows
    want = listed + wt[:, D - 1] * (S64[:n_own].sum(0).unsqueeze(0) - plain)
    assert O.rel_err(Y.detach(), want.detach()) <= 1e-5
    dS64, dlut64 = torch.autograd.grad(want, [S64, lut64], up.double())
    assert O.rel_err(dS, dS64) <= 1e-5 and O.rel_err(dlut, dlut64) <= 1e-5


def test_narrow_row_plan_takes_the_low_threshold_only_for_a_short_tail(monkeypatch):
    from gnan_amd import graph as G
    rng = np.random.default_rng(11)
    n = 300
    rowptr, col, code = _csr(n, n, 1, rng, hubs=[(5, 100), (9, 700), (200, 65)])
    g = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(code), n_cols=n, n_codes=3)
    default, narrow = g.long_row_plan(), g.narrow_row_plan()
    assert default.n_long == 1 and default.threshold == G.LONG_ROW_THRESHOLD and int(default.rows[0]) == 9
    assert narrow.n_long == 3 and narrow.threshold == G.LONG_ROW_THRESHOLD_NARROW and narrow.rows.tolist() == [5, 9, 200]
    assert narrow.slice_ptr.tolist() == [0, 1, 2, 3] and g.narrow_row_plan() is narrow and g.long_row_plan() is default
    monkeypatch.setattr(G, "NARROW_PLAN_MAX_ROWS", 2)                   # "many" rows above the low threshold: default plan
    assert g.narrow_row_plan() is default


def test_hot_columns_point_at_the_appended_rows(monkeypatch):
    """HopGraph.hot_columns / degree_sorted_copy_hot (pure index work): the K most listed neighbours, most listed first,
    ties by id; every pair that lists one of them points at n_cols + rank in the copy, all other ids are untouched;
    flat or small graphs get no hot set."""
    from gnan_amd import HopGraph, graph as G
    from gnan_amd.aggregate import append_hot_rows
    monkeypatch.setattr(G, "HOT_COLUMNS", 8)
    monkeypatch.setattr(G, "HOT_COLUMNS_MIN_NNZ", 0)
    rng = np.random.default_rng(3)
    n = 400
    deg = rng.integers(0, 9, n)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    col = rng.integers(0, n, nnz)
    hot = rng.random(nnz) < 0.5
    col[hot] = rng.integers(0, 8, int(hot.sum())) * 31 + 5
    code = rng.integers(0, 2, nnz).astype(np.uint8)
    g = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col).int(), torch.from_numpy(code), n_cols=n, n_codes=3)
    ids = g.hot_columns()
    listed = np.bincount(col, minlength=n)
    want = np.lexsort((np.arange(n), -listed))[:8]
    assert ids.tolist() == want.tolist()
    copy, order, hot_ids = g.degree_sorted_copy_hot()
    plain, order2, _ = g.degree_sorted_copy()
    assert hot_ids is ids and torch.equal(order, order2) and copy.n_cols == n + 8
    assert torch.equal(copy.rowptr, plain.rowptr) and torch.equal(copy.code, plain.code)
    rank = {int(v): k for k, v in enumerate(want)}
    expect = [n + rank[c] if c in rank else c for c in plain.col.tolist()]
    assert copy.col.tolist() == expect
    S = torch.arange(n * 2 * 3, dtype=torch.float32).view(n * 2, 3)        # two rows per node (s_by_code layout)
    ext = append_hot_rows(S, ids, 2)
    assert ext.shape == ((n + 8) * 2, 3) and torch.equal(ext[: n * 2], S)
    assert torch.equal(ext[n * 2:].view(8, 6), S.view(n, 6)[ids])
    flat = HopGraph.from_csr(torch.arange(0, 4 * n + 1, 4), torch.arange(4 * n).int() % n, torch.zeros(4 * n, dtype=torch.uint8),
                             n_cols=n, n_codes=3)
    assert flat.hot_columns() is None and flat.degree_sorted_copy_hot()[2] is None


def test_bench_refuses_to_time_fewer_ranks_than_asked():
    """``python bench.py --gpus 2`` with fewer than two visible GPUs exits non-zero instead of timing one rank and labelling
    it two (the parent counts devices without initialising one and starts no ranks)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()
    # a launcher's world size that disagrees with --gpus is an error too, before any device work
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "2", "0", "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
