"""The bucketed copy of the adjacency behind ``gnan_spmm_pb_fwd`` (``HopGraph.pb_plan``) — CPU checks of the index work.

``emulate`` restates what the two kernels of csrc/spmm_pb.hip do with the plan's arrays (numpy, float64; TEST INFRASTRUCTURE:
the GPU suite runs the kernels themselves, tests/test_gpu_kernels.py); its result must be the oracle's aggregation."""
import numpy as np
import pytest
import torch

import gnan_amd  # noqa: F401
from gnan_amd import HopGraph, graph as G
from oracle import gnan_oracle as O


def random_graph(rng, n_rows, n_cols, D, hubs=(), self_pairs=True, mean_deg=6):
    """Hop-coded CSR: every row's self pair with code 0 (when ``self_pairs``), other pairs with codes 1..D-2."""
    deg = rng.poisson(mean_deg, n_rows)
    deg[rng.random(n_rows) < 0.1] = 0
    for r, d in hubs:
        deg[r] = d
    cols, codes = [], []
    for i in range(n_rows):
        c = rng.integers(0, n_cols, deg[i])
        k = rng.integers(1, max(2, D - 1), deg[i]) if D > 2 else np.zeros(deg[i], dtype=np.int64)
        if self_pairs and D > 2:
            c, k = np.concatenate([[i % n_cols], c]), np.concatenate([[0], k])
        cols.append(c)
        codes.append(k)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum([len(c) for c in cols])
    col = np.concatenate(cols).astype(np.int32)
    code = np.concatenate(codes).astype(np.uint8)
    g = HopGraph.from_csr(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(code), n_cols=n_cols, n_codes=D)
    return g, rowptr, col, code


def emulate(plan, g, S, lut, use_cnt, s_total):
    """pb_expand_kernel + pb_reduce_kernel on the plan's arrays."""
    W = S.shape[1]
    S = S.double().numpy()
    src, dst = plan.src.numpy().astype(np.int64), plan.dst.numpy().astype(np.int64)
    E = np.full((plan.n_entries, W), np.nan)
    written = np.zeros(plan.n_entries, dtype=np.int64)
    chunk_q, cptr = plan.chunk_q.numpy().astype(np.int64), plan.cb_chunk_ptr.numpy()
    for cb in range(plan.n_cblocks):
        q = (chunk_q[cptr[cb]:cptr[cb + 1], None] + np.arange(G.PB_CHUNK)[None, :]).ravel()
        rows = cb * plan.cb_width + src[q]
        assert np.all(src[q] < plan.cb_width)
        E[q] = S[np.minimum(rows, g.n_cols - 1)]               # pads read row 0 of the block (any finite value will do)
        written[q] += 1
    assert np.all(written == 1), "every entry is expanded by exactly one column block"
    n, D = g.n_rows, g.n_codes
    out = np.zeros((n, W))
    cnt = np.maximum(g.cnt.numpy(), 1).astype(np.float64)
    lutv = lut.double().numpy().reshape(-1)
    eptr, rptr, slot_ptr = plan.bin_entry_ptr.numpy(), plan.bin_row_ptr.numpy(), plan.slot_ptr.numpy()
    seen_rows = np.zeros(n, dtype=np.int64)
    for b in plan.bin_order.numpy():
        acc = np.zeros((plan.acc_per_bin, W))
        q = np.arange(eptr[b], eptr[b + 1])
        assert np.all(dst[q] < plan.acc_per_bin)
        np.add.at(acc, dst[q], E[q])
        slot0 = slot_ptr[rptr[b]]
        assert slot_ptr[rptr[b + 1]] - slot0 <= (plan.acc_per_bin - 1) // plan.n_acc
        for i in range(rptr[b], rptr[b + 1]):
            seen_rows[i] += 1
            wt = lutv / cnt[i] if use_cnt else lutv
            w_rest = wt[D - 1] if s_total is not None else 0.0
            y = np.zeros(W)
            if plan.self_col is not None and plan.self_col[i] >= 0:
                y += (wt[0] - w_rest) * S[int(plan.self_col[i])]
            for a in range(plan.n_acc):
                t = sum(acc[(s - slot0) * plan.n_acc + a] for s in range(slot_ptr[i], slot_ptr[i + 1]))
                y += (wt[plan.code_base + a] - w_rest) * t
            if s_total is not None:
                y += w_rest * s_total.double().numpy()
            out[i] = y
    assert np.all(seen_rows == 1), "every row belongs to exactly one bin"
    return torch.from_numpy(out)


@pytest.mark.parametrize("D,W,self_pairs,use_cnt,with_rest", [(3, 1, True, True, True), (3, 2, True, False, True),
                                                               (4, 1, True, True, False), (4, 4, False, True, True),
                                                               (3, 1, False, True, True), (2, 1, False, True, True)])
def test_plan_reproduces_the_aggregation(D, W, self_pairs, use_cnt, with_rest, monkeypatch):
    """Small LDS budget => many bins and column blocks on a small graph; hub rows own several accumulator slots."""
    monkeypatch.setattr(G, "PB_LDS_BYTES", 1024 * W)       # 128 accumulators per bin, 256 operand rows per block
    monkeypatch.setattr(G, "PB_SLOT_PAIRS", 8)
    rng = np.random.default_rng(D * 10 + W)
    n_rows, n_cols = 700, 900
    g, rowptr, col, code = random_graph(rng, n_rows, n_cols, D, hubs=[(5, 60), (333, 150)], self_pairs=self_pairs)
    plan = g.pb_plan(W)
    assert plan is not None and g.pb_plan(W) is plan                                 # cached
    assert plan.n_bins > 3 and plan.n_cblocks > 3 and plan.n_entries % G.PB_CHUNK == 0
    if self_pairs and D > 2:
        assert plan.code_base == 1 and plan.n_acc == D - 2 and plan.n_pairs == g.nnz - n_rows
        assert torch.equal(plan.self_col.long(), torch.arange(n_rows) % n_cols)
    elif D == 2:                                                                     # codes {0, rest}: code 0 is all there is
        assert plan.code_base == 0 and plan.n_acc == 1 and plan.self_col is None
    pads = plan.dst.long() == plan.acc_per_bin - 1
    assert int((~pads).sum()) == plan.n_pairs
    sizes = (plan.bin_entry_ptr[1:] - plan.bin_entry_ptr[:-1])[plan.bin_order.long()]
    assert bool((sizes[1:] <= sizes[:-1]).all())                                     # largest bins first
    S = torch.from_numpy(rng.standard_normal((n_cols, W)).astype(np.float32))
    lut = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32))
    total = S.double().sum(0) if with_rest else None
    got = emulate(plan, g, S, lut, use_cnt, total)
    wt = O.weight_table(lut.double(), g.cnt.long().numpy() if use_cnt else None).expand(n_rows, -1, -1)
    want = O.spmm_csr(rowptr, col, code, S.double(), wt, with_rest=with_rest)
    assert float((got - want).abs().max()) <= 1e-12 * max(1.0, float(want.abs().max()))


def test_plan_declines_what_the_kernels_cannot_take(monkeypatch):
    rng = np.random.default_rng(0)
    g, *_ = random_graph(rng, 50, 60, 3)
    assert g.pb_plan(3) is None                                   # widths other than 1, 2, 4
    dense = HopGraph(n_rows=4, n_cols=4, n_codes=3, code=torch.zeros((4, 4), dtype=torch.uint8),
                     cnt=torch.ones((4, 3), dtype=torch.int32))
    assert dense.pb_plan(1) is None
    only_self, *_ = random_graph(rng, 50, 60, 3, mean_deg=0)       # nothing but self pairs: nothing to bucket
    assert only_self.pb_plan(1) is None
    monkeypatch.setattr(G, "PB_LDS_BYTES", 256)                    # 31 slots per bin; a 400-pair row needs 50 of 8 pairs
    monkeypatch.setattr(G, "PB_SLOT_PAIRS", 8)
    big, *_ = random_graph(rng, 50, 60, 3, hubs=[(7, 400)])
    assert big.pb_plan(1) is None


def test_two_code_zero_pairs_in_a_row_stay_in_the_tiles():
    """The self-pair shortcut needs at most one code-0 pair per row; otherwise code 0 is bucketed like any other code."""
    rowptr = torch.tensor([0, 3, 5])
    col = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32)
    code = torch.tensor([0, 0, 1, 1, 0], dtype=torch.uint8)
    g = HopGraph.from_csr(rowptr, col, code, n_cols=2, n_codes=3)
    plan = g.pb_plan(1)
    assert plan.self_col is None and plan.code_base == 0 and plan.n_acc == 2 and plan.n_pairs == 5
    S = torch.tensor([[2.0], [5.0]])
    lut = torch.tensor([[1.0], [10.0], [100.0]])
    got = emulate(plan, g, S, lut, False, None)
    assert got.flatten().tolist() == [2 + 5 + 50, 20 + 5]
