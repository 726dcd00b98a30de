"""Plain-torch stand-ins for the ``gnan_spmm_*`` launchers — TEST INFRASTRUCTURE ONLY.

They let the CPU suite drive the product's host logic around the kernels (``aggregate._RhoAggregate``: which
launches a backward pass makes, with which operands, and what it adds on top — the rest-bucket terms, the
collectives of the multi-rank variants) without a GPU: ``install()`` swaps them in for
``aggregate.spmm_launch`` / ``shell_sums_launch`` / ``lut_grad_launch`` / ``functional.column_sums`` inside the calling (test)
process.  Each restates the documented semantics of its launcher (aggregate.py, include/gnan_hip.h) with index
arithmetic in float64; the GPU suite checks the real kernels against the oracle, this file is checked against the
oracle in tests/test_host_logic.py.  CSR graphs only.
"""
import torch


def _pairs(g):
    rowptr = g.rowptr.long()
    deg = rowptr[1:] - rowptr[:-1]
    row_of_pair = torch.repeat_interleave(torch.arange(g.n_rows), deg)
    return row_of_pair, g.col.long(), g.code.long()


def column_sums(S):
    return S.detach().float().sum(0)


def spmm_launch(g, S, lut, use_cnt, with_rest, row_ids=None, weight_by_col=False, minus_rest=False, s_total=None,
                reduce_cr=0, s_by_code=False, lut_of_counts=None, lut_channels=1, room=None, keep_shell=None):
    assert not g.is_dense
    if lut is None:
        lut = lut_of_counts(g.cnt)
    S, lut = S.detach().double(), lut.detach().double()
    D, Cw = lut.shape[-2], lut.shape[-1]
    per_row = lut.dim() == 3
    W = S.shape[1]
    row_of_pair, col, code = _pairs(g)
    cnt = g.cnt.clamp_min(1).double()
    owner = col if weight_by_col else row_of_pair               # whose table row weights the pair
    w = lut[owner, code] if per_row else lut[code]               # [nnz, Cw]
    if use_cnt:
        w = w / cnt[owner, code].unsqueeze(-1)
    if minus_rest:
        wr = lut[owner, D - 1] if per_row else lut[D - 1].expand(col.numel(), Cw)
        if use_cnt:
            wr = wr / cnt[owner, D - 1].unsqueeze(-1)
        w = w - wr
    src = S[col * D + code] if s_by_code else S[col]
    Y = torch.zeros((g.n_rows, W), dtype=torch.float64).index_add_(0, row_of_pair, w.repeat(1, W // Cw) * src)
    if with_rest:
        total = S.sum(0) if s_total is None else s_total.detach().double()
        listed = torch.zeros((g.n_rows, W), dtype=torch.float64).index_add_(0, row_of_pair, src)
        w_rest = lut[:, D - 1] if per_row else lut[D - 1].expand(g.n_rows, Cw)
        if use_cnt:
            w_rest = w_rest / cnt[:, D - 1].unsqueeze(-1)
        Y = Y + w_rest.repeat(1, W // Cw) * (total.unsqueeze(0) - listed)
    if row_ids is not None:
        Y = Y[row_ids.long()]
    if reduce_cr:
        Y = Y.view(Y.shape[0], -1, reduce_cr).sum(1)
    return Y.float()


def shell_sums_launch(g, S, lut_like, with_rest, row_ids=None, s_total=None):
    S = S.detach().double()
    D, W = g.n_codes, S.shape[1]
    row_of_pair, col, code = _pairs(g)
    T = torch.zeros((g.n_rows * D, W), dtype=torch.float64).index_add_(0, row_of_pair * D + code, S[col]).view(g.n_rows, D, W)
    if with_rest:
        total = S.sum(0) if s_total is None else s_total.detach().double()
        T[:, D - 1] = total.unsqueeze(0) - T[:, : D - 1].sum(1)
    if row_ids is not None:
        T = T[row_ids.long()]
    return T.float()


def lut_grad_launch(g, S, dY, D, use_cnt, with_rest, row_ids, s_total, reduce_rows):
    T = shell_sums_launch(g, S, None, with_rest, row_ids, s_total).double()          # [n_out, D, W]
    dY = dY.detach().double()
    W = T.shape[2]
    dwt = (T * dY.repeat(1, W // dY.shape[1]).unsqueeze(1)).sum(2)                   # [n_out, D]
    if use_cnt:
        cnt = g.cnt if row_ids is None else g.cnt[row_ids.long()]
        dwt = dwt / cnt.clamp_min(1).double()
    out = dwt.sum(0) if reduce_rows else dwt
    return out.float().unsqueeze(-1)


def pack_bwd_rows(dY, cnt, D, with_rest, half, hot=None, want_q_sum=False):
    assert hot is None
    dY = dY.detach().float()
    n, W = dY.shape
    den = torch.ones((n, D)) if cnt is None else cnt.clamp_min(1).float()
    V = torch.zeros((n, D, 2 * half))
    V[:, :, :W] = dY.unsqueeze(1) / den.unsqueeze(-1)
    if with_rest:
        V[:, :, half:half + W] = (dY / den[:, D - 1:D]).unsqueeze(1)
    V = V.permute(1, 0, 2).contiguous()                         # code-major [D, n, 2 * half]
    if want_q_sum:
        return V, V[0, :, half:half + 1].double().sum(0).float()
    return V


def bwd_narrow_launch(gt, V, S_rows, lut, with_rest, W, walk=None, ds_add=None, rest_q=None, rest_total=None, add_to_rows=False):
    D = lut.numel()
    half = V.shape[1] // 2
    row_of_pair, col, code = _pairs(gt)                     # rows of gt = nodes as neighbours; col = the forward row listing them
    Vd = V.detach().double()[code * gt.n_cols + col]           # code-major rows
    n = gt.n_rows
    A = torch.zeros((n * D, half), dtype=torch.float64).index_add_(0, row_of_pair * D + code, Vd[:, :half]).view(n, D, half)[:, :, :W]
    Q = torch.zeros((n, half), dtype=torch.float64).index_add_(0, row_of_pair, Vd[:, half:])[:, :W]
    l, Sd, rest = lut.detach().double().reshape(-1), S_rows.detach().double(), D - 1
    ds, dl = torch.zeros((n, W), dtype=torch.float64), torch.zeros(D, dtype=torch.float64)
    for d in range(D):
        if d < rest or not with_rest:
            ds += l[d] * A[:, d]
            dl[d] = (Sd * A[:, d]).sum()
    if with_rest:
        ds -= l[rest] * Q
        dl[rest] = -(Sd * Q).sum()
    if ds_add is not None:
        ds += ds_add.detach().double().reshape(1, -1)
    if rest_q is not None and add_to_rows:
        ds += (l[rest].float() * rest_q.detach().float()).double().reshape(1, -1)
    if rest_q is not None and rest_total is not None:
        dl[rest] += (rest_total.detach().double().reshape(-1) * rest_q.detach().double().reshape(-1)).sum()
    return ds.float(), dl.float()


def install():
    """Swap the stand-ins in (call inside the test process / spawned worker; undo with the returned function)."""
    from gnan_amd import _lib, aggregate, functional
    mine = {"spmm_launch": spmm_launch, "shell_sums_launch": shell_sums_launch, "lut_grad_launch": lut_grad_launch,
            "bwd_narrow_launch": bwd_narrow_launch, "pack_bwd_rows": pack_bwd_rows}
    saved = {name: getattr(aggregate, name) for name in mine}
    saved_sums, saved_req = functional.column_sums, _lib.require_device
    for name, fn in mine.items():
        setattr(aggregate, name, fn)
    functional.column_sums = column_sums                       # (aggregate reads it through the functional module)
    _lib.require_device = lambda *a, **k: None

    def undo():
        for name, fn in saved.items():
            setattr(aggregate, name, fn)
        functional.column_sums = saved_sums
        _lib.require_device = saved_req
    return undo


def rest_total_term(g, lut, use_cnt, total, reduce_channels=0, row_ids=None):
    """The part of the aggregation that depends on the operand only through its column sums:
    ``R[q, w] = wt(i_q, D-1, w) * total[w]`` (summed per channel ``w mod reduce_channels`` with the fused read-out).

    ``rho_aggregate(..., s_total=total) == rho_aggregate(..., s_total=zeros) + rest_total_term(..., total)`` up to
    rounding: a multi-rank forward can run the aggregation while the all-reduce of ``total`` is still in flight and add
    this term afterwards (inference only: no autograd through it)."""
    D, Cw = lut.shape[-2], lut.shape[-1]
    W = int(total.numel())
    rows = None if row_ids is None else row_ids.long()
    with torch.no_grad():
        if lut.dim() == 2 and Cw == 1 and rows is None and (reduce_channels or W == 1):
            # the common case in three small launches: per-channel sums of total, times rho(0), times the cached 1/|rest shell|
            t = total.float().view(-1, max(reduce_channels, 1)).sum(0) * lut[D - 1, 0].float()
            inv = g.inv_rest_count() if use_cnt else torch.ones((g.n_rows, 1), dtype=torch.float32, device=total.device)
            return inv * t.unsqueeze(0)
        if lut.dim() == 3:
            w_rest = (lut[:, D - 1, :] if rows is None else lut[rows, D - 1, :]).float()          # [n, Cw]
        else:
            w_rest = lut[D - 1].float().unsqueeze(0)                                              # [1, Cw]
        if use_cnt:
            c = g.cnt[:, D - 1:D] if rows is None else g.cnt[rows, D - 1:D]
            w_rest = w_rest / c.clamp_min(1).float()
        n = g.n_rows if rows is None else int(rows.numel())
        w_rest = w_rest.expand(n, Cw)
        idx = torch.arange(W, device=total.device)
        if reduce_channels:
            A = torch.zeros((Cw, reduce_channels), dtype=torch.float32, device=total.device)
            A.index_put_((idx % Cw, idx % reduce_channels), total.float(), accumulate=True)
            return w_rest @ A
        return w_rest[:, idx % Cw] * total.float().unsqueeze(0)
