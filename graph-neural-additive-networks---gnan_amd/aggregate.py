"""The rho(distance)-weighted, shell-normalised neighbourhood sum (GNAN.py:65-73 / models.py:368-376) and its backward pass:
launch wrappers of ``csrc/spmm.hip`` (``spmm_launch``, ``shell_sums_launch``, ``lut_grad_launch``, ``bwd_narrow_launch``,
``pack_bwd_rows``), the dispatch between them (thresholds below), and the autograd nodes ``rho_aggregate`` /
``pre_rho_aggregate`` / ``reference_order_forward`` are built from.  Shape functions live in ``functional``; the two meet
in ``modules``.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from . import functional as Fn
from .functional import StackedMLP
from .graph import HopGraph

DENSE_SLICE_MAX_ROWS = 16384  # dense layout: slice every row over workgroups while row blocks alone would not fill the GPU
DENSE_SLICE_MIN_COLS = 512
WEIGHT_TABLE_MAX_BYTES = 1 << 30   # backward w.r.t. a wide S: per-node table of the pairs' weights while it stays below 1 GiB
NARROW_DS_MAX_WIDTH = 32     # backward w.r.t. S: pre-weighted (node, hop code) operand while its rows stay <= 128 B


def _spmm_args(g: HopGraph, S, lut, use_cnt, s_total, out, row_ids, per_row_lut, weight_by_col=False,
               minus_rest=False, plan=None, workspace=None, reduce_cr=0, scatter_out=False,
               s_by_code=False, packed=False, hot_rows=0) -> _lib.SpmmArgs:
    D, Cw = lut.shape[-2], lut.shape[-1]
    # one index stream (col | code << 29) where the graph carries it and the kernel variant reads it (gnan_hip.h)
    packed = bool(packed and PACKED_INDEX and g.colp is not None and D <= 4 and Cw == 1 and not weight_by_col
                  and not minus_rest and not s_by_code)
    a = _lib.SpmmArgs(
        n_rows=out.shape[0], n_cols=g.n_cols,
        rowptr=_lib.ptr(g.rowptr), rowptr_is64=int(g.rowptr is not None and g.rowptr.dtype == torch.int64),
        col=_lib.ptr(g.colp if packed else g.col), code=_lib.ptr(g.code), row_ids=_lib.ptr(row_ids),
        S=_lib.ptr(S), s_dtype=_lib.GNAN_BF16 if S.dtype == torch.bfloat16 else _lib.GNAN_F32, W=S.shape[1],
        s_stride=S.stride(0),
        lut=_lib.ptr(lut), lut_row_stride=(D * Cw if per_row_lut else 0), D=D, Cw=Cw,
        cnt=_lib.ptr(g.cnt) if use_cnt else None, cnt_stride=g.cnt.stride(0),
        s_total=_lib.ptr(s_total), weight_by_col=int(weight_by_col), minus_rest=int(minus_rest),
        reduce_cr=int(reduce_cr), scatter_out=int(scatter_out), Y=_lib.ptr(out), y_stride=out.stride(0),
        long_threshold=(plan.threshold if plan is not None else 0),
        long_rows=_lib.ptr(plan.rows) if plan is not None else None,
        long_slice_ptr=_lib.ptr(plan.slice_ptr) if plan is not None else None,
        n_long=(plan.n_long if plan is not None else 0), n_slices=(plan.n_slices if plan is not None else 0),
        slice_edges=(plan.slice_edges if plan is not None else 0),
        workspace=_lib.ptr(workspace), workspace_bytes=(workspace.numel() * 4 if workspace is not None else 0),
        s_by_code=int(s_by_code), nnz=(0 if (g.col is None or not WIDE_INDEX_LOADS) else int(g.col.numel())),
        packed_index=int(packed))
    if hot_rows and packed and HOT_ROWS_IN_LDS:
        # the appended compact copy of the most listed rows sits behind the real ones: its head is served from LDS — where it
        # receives enough of the pairs to pay for the persistent kernel's lower occupancy (10M-node R-MAT: 32 % at W = 1,
        # 1.04 -> 0.91 ms; the 111M-node graph: 20.1 -> 20.7 ms, so not there)
        W = S.shape[1]
        head = min(int(hot_rows), HOT_LDS_FLOATS // max(W, 1))
        share = (getattr(g, "_hot_head_share", None) or {}).get(head, 0.0)
        if share >= HOT_LDS_MIN_SHARE:
            a.hot_lo, a.hot_rows = g.n_cols - int(hot_rows), head
    return a


FUSABLE_READOUT = (1, 2, 4)   # channel counts the aggregation kernel can sum over features in its epilogue
DEGREE_SCHEDULE_MIN_WIDTH = 8  # operand widths from which the degree-sorted row schedule pays (measured: W >= 8)
DEGREE_SORTED_COPY_MIN_ROWS = 1 << 16   # below this the copy's one-off index work outweighs what the kernel saves
NARROW_ROW_SLICING = True   # A/B switch of LONG_ROW_THRESHOLD_NARROW
WIDE_INDEX_LOADS = True      # a lane's run of index entries as 16-byte loads (gnan_spmm_args.nnz)
PACKED_INDEX = True         # degree-sorted copies are read as one (col | code << 29) stream
NARROW_SORTED_MIN_NNZ = 1 << 23   # below ~8M pairs the sorted walk's tail (the longest rows run last) and its scattered stores cost
                                   # more than the divergence they remove (arxiv-shaped, 1.3M pairs: 11.6 -> 26 us at W = 1)
NARROW_SORTED_WALK = True   # narrow operand rows walk the degree-sorted copy too
HOT_COLUMN_ROWS = True        # ... and read the most listed neighbours from a compact copy
HOT_ROWS_IN_LDS = True           # ... and serve the head of that copy from LDS (spmm_hot_kernel)
HOT_LDS_FLOATS = 16384                                                   # 64 KB per workgroup, two workgroups per CU
HOT_LDS_MIN_SHARE = 0.25                                                 # ... from this share of the pairs listing the LDS-resident rows
DEGREE_SORTED_COPY = True      # ... through a degree-sorted copy of the CSR (HopGraph.degree_sorted_copy) instead of an index


def append_hot_rows(S: torch.Tensor, hot: torch.Tensor, group: int = 1, room: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``S`` ([n * group, W], ``group`` consecutive rows per node) followed by the rows of the nodes ``hot``: the operand of
    a graph whose column ids point hot neighbours at ``n + rank`` (``HopGraph.hot_columns``).  ``room``: a buffer whose first
    rows ARE ``S`` with space for the copy behind them (``feature_mlps(room_rows=...)``): the copy is gathered in place."""
    W = S.shape[1]
    n = S.shape[0] // group
    if (room is not None and group == 1 and room.data_ptr() == S.data_ptr() and room.shape[1] == W and room.dtype == S.dtype
            and room.is_contiguous() and S.is_contiguous() and room.shape[0] >= n + hot.numel()):
        # written through the library, not through torch: S is an autograd Function's output heading `room`, and a tracked
        # in-place write into its base would (rightly, in general) invalidate it for the backward pass — these rows are not S
        h = hot.to(torch.int64).contiguous()
        _lib.check(_lib.lib().gnan_gather_rows(_lib.ptr(S), S.stride(0), _lib.ptr(h), int(h.numel()), W,
                                               room.data_ptr() + n * W * 4, _lib.stream_of(S)), "gnan_gather_rows")
        return room[: n + hot.numel()]
    ext = torch.empty(((n + hot.numel()) * group, W), dtype=S.dtype, device=S.device)
    ext[: n * group].copy_(S)
    torch.index_select(S.contiguous().view(n, group * W), 0, hot, out=ext[n * group:].view(hot.numel(), group * W))
    return ext


def spmm_launch(g: HopGraph, S: torch.Tensor, lut: Optional[torch.Tensor], use_cnt: bool, with_rest: bool,
                row_ids: Optional[torch.Tensor] = None, weight_by_col: bool = False,
                minus_rest: bool = False, s_total: Optional[torch.Tensor] = None, reduce_cr: int = 0,
                s_by_code: bool = False, lut_of_counts=None, lut_channels: int = 1, room=None,
                keep_shell: Optional[list] = None) -> torch.Tensor:
    """One ``gnan_spmm_fwd`` call (no autograd).  ``lut`` is ``[D, Cw]`` or ``[n_adj_rows, D, Cw]``.
    ``reduce_cr`` in FUSABLE_READOUT returns ``[n, reduce_cr]`` = per-channel sums over the operand columns.
    ``s_by_code``: ``S`` is ``[n_cols * D, W]`` and the pair with neighbour ``c`` and hop code ``d`` reads row ``c*D + d``.
    ``lut_of_counts`` (with ``lut=None``): a function ``cnt [n, D] -> [n, D, lut_channels]`` giving the per-row table of a
    graph from its shell counts (the pre-rho normalisation, :func:`pre_rho_aggregate`); it is called on the counts of the
    graph that is actually walked, so a degree-sorted copy gets its table in its own row order and nothing is permuted."""
    _lib.require_device(S, lut, g.code)
    whole = row_ids is None                              # (the walks below may bring row_ids of their own)
    S = S.detach()
    if S.dtype != torch.bfloat16:                        # bf16 rows: storage format only, accumulation stays fp32
        S = S.float()
    S = Fn._rows(S)
    from_counts = lut is None
    if from_counts:
        if lut_of_counts is None:
            raise ValueError("spmm_launch needs a weight table or a function of the shell counts")
        lut_shape, per_row = (g.n_codes, lut_channels), False       # (per_row: a table the walk's order does not constrain)
    else:
        lut = lut.detach().float().contiguous()
        per_row = lut.dim() == 3
        lut_shape = (lut.shape[-2], lut.shape[-1])
    if lut_shape[0] != g.n_codes:
        raise ValueError(f"weight table has {lut_shape[0]} codes, graph has {g.n_codes}")
    if S.shape[0] != g.n_cols * (g.n_codes if s_by_code else 1):
        raise ValueError(f"operand has {S.shape[0]} rows, graph has {g.n_cols} neighbour nodes")
    if S.shape[1] % lut_shape[1] != 0:
        raise ValueError("operand width must be a multiple of the weight-channel count")
    n_out = g.n_rows if row_ids is None else int(row_ids.numel())
    out = torch.empty((n_out, reduce_cr if reduce_cr else S.shape[1]), dtype=torch.float32, device=S.device)
    if with_rest and s_total is None:
        s_total = Fn.column_sums(S)
    if not with_rest:
        s_total = None
    scatter = False
    n_hot = 0
    narrow = S.shape[1] * S.element_size() <= 8        # one or two lanes per row: see LONG_ROW_THRESHOLD_NARROW
    if (PB_NARROW and not g.is_dense and row_ids is None and S.dtype == torch.float32 and S.shape[1] in PB_WIDTHS and not per_row
            and not from_counts and lut_shape[1] == 1 and lut_shape[0] <= 4 and not weight_by_col and not minus_rest
            and not s_by_code and reduce_cr == 0 and PB_MIN_NNZ <= g.nnz <= PB_MAX_NNZ and not g._cnt_by_col):
        # narrow rows of a large graph: no per-pair gather at all — bucketed pairs, operand blocks and accumulators in LDS
        # (csrc/spmm_pb.hip; 10M-node R-MAT, W = 1: 0.87 ms of spmm_hot_kernel -> see DESIGN.md section 4.1c)
        pb = g.pb_plan(S.shape[1])
        if pb is not None:
            shell = None
            if keep_shell is not None and PB_BACKWARD_ONE_COLUMN and S.shape[1] == 1 and pb.n_acc == 1:
                # a training forward keeps the rows' raw shell sums (4 bytes per row): its backward then needs no second column
                shell = torch.empty(g.n_rows, dtype=torch.float32, device=S.device)
                keep_shell.append(shell)
            return pb_launch(g, pb, S, lut, use_cnt, s_total, out, shell_out=shell)
    if g.is_dense:
        # one lane group per row fills the chip only with >~ 16k rows; below that every row is sliced over workgroups
        plan = g.dense_slice_plan(n_out) if (n_out < DENSE_SLICE_MAX_ROWS and g.n_cols >= DENSE_SLICE_MIN_COLS) else None
    elif (NARROW_SORTED_WALK and DEGREE_SORTED_COPY and row_ids is None and S.shape[1] < DEGREE_SCHEDULE_MIN_WIDTH
          and S.dtype == torch.float32 and not per_row and not weight_by_col and g.n_rows >= DEGREE_SORTED_COPY_MIN_ROWS
          and g.nnz >= NARROW_SORTED_MIN_NNZ):
        # narrow operand rows: the degree-sorted copy as well (a lane per row idles behind the longest of the 16..64 rows
        # of its wavefront: W = 2 on the 10M-node graph 2.68 -> 1.68 ms), and a compact copy of the most listed
        # neighbours' rows behind the operand (HopGraph.hot_columns: -> 1.33 ms; W = 1: 1.86 -> 1.79 -> 1.52 ms)
        g, row_ids, hot = narrow_walk(g)
        plan = g.narrow_row_plan() if (narrow and NARROW_ROW_SLICING) else g.long_row_plan()
        if hot is not None:
            S = append_hot_rows(S, hot, g.n_codes if s_by_code else 1, room=room)
            n_hot = 0 if s_by_code else int(hot.numel())
        scatter = 2
    elif row_ids is None and S.shape[1] >= DEGREE_SCHEDULE_MIN_WIDTH and g.n_rows > 1:
        # (a table indexed by COLUMN — the wide backward's per-node weights — does not care in which order the rows are walked)
        by_col_table = per_row and weight_by_col and not use_cnt
        if DEGREE_SORTED_COPY and (by_col_table or (not per_row and not weight_by_col)) and g.n_rows >= DEGREE_SORTED_COPY_MIN_ROWS:
            g, row_ids, plan = g.degree_sorted_copy()   # walk a degree-sorted copy of the CSR, store rows at their own index
            scatter = 2
        else:
            row_ids, plan = g.degree_schedule()      # process rows by degree, store them in place
            scatter = True
    else:
        plan = g.narrow_row_plan() if (narrow and NARROW_ROW_SLICING and row_ids is None) else g.long_row_plan(row_ids)
    if from_counts:
        lut = lut_of_counts(g.cnt).detach().float().contiguous()          # rows of THIS graph (a sorted copy carries its own counts)
        per_row = True
    a = _spmm_args(g, S, lut, use_cnt, s_total, out, row_ids, per_row, weight_by_col, minus_rest, plan,
                   reduce_cr=reduce_cr, scatter_out=scatter, s_by_code=s_by_code, packed=True, hot_rows=n_hot)
    need = _lib.lib().gnan_spmm_fwd_workspace_bytes(a)
    ws = None
    if need:
        ws = torch.empty(need // 4, dtype=torch.float32, device=S.device)
        a.workspace, a.workspace_bytes = _lib.ptr(ws), need
    if (keep_shell is not None and ROWS_BACKWARD_ONE_COLUMN and S.shape[1] == 1 and S.dtype == torch.float32 and not g.is_dense
            and not per_row and lut_shape[1] == 1 and lut_shape[0] <= 4 and not weight_by_col and not minus_rest and not s_by_code
            and reduce_cr == 0 and n_hot == 0 and a.n_slices == 0 and whole and not g._cnt_by_col):
        # a training forward of a one-column operand keeps its rows' raw per-code sums ([n, D - 1]): the backward is then a pass
        # over rows plus one gather of pre-weighted numbers (rows_bwd1_launch) instead of two numbers per row over the transposed pairs
        shell = torch.empty((n_out, lut_shape[0] - 1), dtype=torch.float32, device=S.device)
        a.shell_out = _lib.ptr(shell)
        keep_shell.append(shell)
    _lib.check(_lib.lib().gnan_spmm_fwd(a, _lib.stream_of(S)), "gnan_spmm_fwd")
    return out


_ONES = {}


def _ones_table(dev, D: int) -> torch.Tensor:
    """``[D, 1]`` ones on ``dev`` (a constant: made once, not by a fill launch per backward)."""
    t = _ONES.get((dev, D))
    if t is None:
        t = torch.ones((D, 1), dtype=torch.float32, device=dev)
        if not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):      # (a capture's allocations are its own)
            _ONES[(dev, D)] = t
    return t


ROWS_BACKWARD_ONE_COLUMN = True     # row-parallel route: keep the forward's per-code shell sums, backward by gnan_spmm_pack_z


def rows_bwd1_launch(g: HopGraph, dY: torch.Tensor, shell: torch.Tensor, lut: torch.Tensor, use_cnt: bool, with_rest: bool,
                     s_total: Optional[torch.Tensor]):
    """``(dS [n_cols, 1], dlut [D])`` of the one-column aggregation on the row-parallel route from the forward's kept per-code shell
    sums ``[n, D - 1]``: ``gnan_spmm_pack_z`` (a pass over the rows: Z, q, the table gradient), then the forward kernel over the
    transposed adjacency with one pre-weighted operand row per (node, hop code)."""
    _lib.require_device(dY, shell, lut)
    n, dev = g.n_rows, dY.device
    D = int(lut.numel())
    dY = Fn._rows(dY.detach().float())
    lut = lut.detach().float().reshape(-1).contiguous()
    Z = torch.empty((n * D, 1), dtype=torch.float32, device=dev)
    q = torch.empty(1, dtype=torch.float32, device=dev)
    dlut = torch.empty(D, dtype=torch.float32, device=dev)
    need = _lib.lib().gnan_spmm_pack_z_workspace_bytes(n)
    ws = torch.empty((need + 15) // 16 * 2, dtype=torch.float64, device=dev)
    cnt = g.cnt if use_cnt else None
    tot = None if (s_total is None or not with_rest) else s_total.detach().float().reshape(-1).contiguous()
    za = _lib.PackZArgs(n=n, dY=_lib.ptr(dY), dy_stride=dY.stride(0), cnt=_lib.ptr(cnt), cnt_stride=0 if cnt is None else cnt.stride(0),
                        D=D, with_rest=int(with_rest), lut=_lib.ptr(lut), shell=_lib.ptr(shell), s_total=_lib.ptr(tot), Z=_lib.ptr(Z),
                        q=_lib.ptr(q), dlut=_lib.ptr(dlut), workspace=_lib.ptr(ws), workspace_bytes=ws.numel() * 8)
    _lib.check(_lib.lib().gnan_spmm_pack_z(za, _lib.stream_of(dY)), "gnan_spmm_pack_z")
    dS = spmm_launch(g.transposed(), Z, _ones_table(dev, D), False, False, None, s_by_code=True)
    if with_rest:
        dS.addcmul_(lut[D - 1:], q)                          # d/dS_j of wt(i, rest) * total: rho(0) q on every row
    return dS, dlut


PB_NARROW = True            # narrow fp32 rows of large CSR graphs go through the propagation-blocked kernels (gnan_spmm_pb_fwd)
PB_WIDTHS = (1, 2, 4)
PB_FLAGS = 0                # gnan_spmm_pb_args.flags (A/B switches of the kernels)
PB_BACKWARD = True          # ... and so does the one-column backward (gnan_spmm_pb_bwd over the transposed graph's copy)
PB_MIN_NNZ = 1 << 23        # below, the row-parallel kernel's gathers stay in L2 and three launches cost more than they save
PB_MAX_NNZ = 1 << 29        # above, building the bucketed copy (a sort of the pairs, ~80 B of temporaries per pair) is not attempted


def pb_launch(g: HopGraph, pb, S: torch.Tensor, lut: torch.Tensor, use_cnt: bool, s_total: Optional[torch.Tensor],
              out: Optional[torch.Tensor] = None, shell_out: Optional[torch.Tensor] = None, S_self: Optional[torch.Tensor] = None,
              out_add=None) -> torch.Tensor:
    """One ``gnan_spmm_pb_fwd`` call over the bucketed copy ``pb = g.pb_plan(W)``: ``[n_rows, W]``, every row, global table.
    ``shell_out [n_rows]`` (W == 1): the rows' raw sums of the accumulated hop code, kept for the backward; ``S_self``: the self
    pairs' operand where it is not ``S``; ``out_add = (value [1], scale [1])``: their product is added to every row."""
    _lib.require_device(S, lut, pb.src)
    W = S.shape[1]
    if S.stride(0) != W or S.stride(1) != 1:
        S = S.contiguous()
    lut = lut.detach().float().reshape(-1).contiguous()
    if out is None:
        out = torch.empty((g.n_rows, W), dtype=torch.float32, device=S.device)
    a = _lib.SpmmPbArgs(n_rows=g.n_rows, n_cols=g.n_cols, S=_lib.ptr(S), s_stride=W, W=W, D=int(lut.numel()), lut=_lib.ptr(lut),
                        cnt=_lib.ptr(g.cnt) if use_cnt else None, cnt_stride=g.cnt.stride(0), s_total=_lib.ptr(s_total),
                        Y=_lib.ptr(out), y_stride=out.stride(0), n_entries=pb.n_entries, src=_lib.ptr(pb.src), dst=_lib.ptr(pb.dst),
                        cb_width=pb.cb_width, n_cblocks=pb.n_cblocks, chunk_q=_lib.ptr(pb.chunk_q),
                        cb_chunk_ptr=_lib.ptr(pb.cb_chunk_ptr), n_bins=pb.n_bins, acc_per_bin=pb.acc_per_bin,
                        bin_order=_lib.ptr(pb.bin_order), bin_entry_ptr=_lib.ptr(pb.bin_entry_ptr),
                        bin_row_ptr=_lib.ptr(pb.bin_row_ptr), slot_ptr=_lib.ptr(pb.slot_ptr), n_acc=pb.n_acc,
                        code_base=pb.code_base, self_col=_lib.ptr(pb.self_col), headroom_bits=pb.headroom_bits, flags=PB_FLAGS, self_is_row=int(pb.self_is_row))
    if shell_out is not None:
        a.shell_out = _lib.ptr(shell_out)
    if S_self is not None:
        a.S_self = _lib.ptr(S_self)
    if out_add is not None:
        a.out_add, a.out_add_scale = _lib.ptr(out_add[0]), _lib.ptr(out_add[1])
    need = _lib.lib().gnan_spmm_pb_workspace_bytes(a)
    ws = torch.empty((need + 15) // 16 * 4, dtype=torch.float32, device=S.device)       # (the caching allocator aligns to 512 B)
    a.workspace, a.workspace_bytes = _lib.ptr(ws), ws.numel() * 4
    _lib.check(_lib.lib().gnan_spmm_pb_fwd(a, _lib.stream_of(S)), "gnan_spmm_pb_fwd")
    return out


PB_BACKWARD_ONE_COLUMN = True   # the one-column backward from the forward's kept shell sums: no second column through the buckets


def pb_bwd1_applies(g: HopGraph, W: int, D: int, with_rest: bool, add_to_rows: bool, shell):
    """``(forward plan, transposed W = 1 plan)`` when the one-column backward can run WITHOUT the packed second column
    (``gnan_spmm_pb_pack1`` + ``gnan_spmm_pb_fwd`` over the transposed adjacency), else None."""
    if not (PB_NARROW and PB_BACKWARD and PB_BACKWARD_ONE_COLUMN and shell is not None and W == 1 and not g.is_dense and D <= 4
            and PB_MIN_NNZ <= g.nnz <= PB_MAX_NNZ and (add_to_rows or not with_rest)):
        return None
    fwd = g.pb_plan(1)
    if fwd is None or fwd.n_acc != 1 or fwd.code_base < 1:
        return None
    pbt = g.transposed().pb_plan(1)
    if pbt is None or pbt.n_acc != 1 or pbt.code_base != fwd.code_base:
        return None
    return fwd, pbt


def pb_bwd1_launch(g: HopGraph, plans, dY: torch.Tensor, S: torch.Tensor, shell: torch.Tensor, lut: torch.Tensor, use_cnt: bool,
                   with_rest: bool, s_total: Optional[torch.Tensor]):
    """``(dS [n_cols, 1], dlut [D])`` of the one-column aggregation: one pass over the ROWS (``gnan_spmm_pb_pack1``: c, e, q and
    the whole table gradient from the kept shell sums), then the forward's two phases over the transposed adjacency with ``c``
    as the operand."""
    fwd, pbt = plans
    _lib.require_device(dY, S, shell, lut)
    n, dev = g.n_rows, dY.device
    D = int(lut.numel())
    dY = Fn._rows(dY.detach().float())
    S = S.detach().float().contiguous()
    lut = lut.detach().float().reshape(-1).contiguous()
    c = torch.empty(n, dtype=torch.float32, device=dev)
    e = torch.empty(n, dtype=torch.float32, device=dev)
    q = torch.empty(1, dtype=torch.float32, device=dev)
    dlut = torch.empty(D, dtype=torch.float32, device=dev)
    need = _lib.lib().gnan_spmm_pb_pack1_workspace_bytes(n)
    ws = torch.empty((need + 15) // 16 * 2, dtype=torch.float64, device=dev)
    cnt = g.cnt if use_cnt else None
    tot = None if (s_total is None or not with_rest) else s_total.detach().float().reshape(-1).contiguous()
    pa = _lib.PbPack1Args(n=n, dY=_lib.ptr(dY), dy_stride=dY.stride(0), cnt=_lib.ptr(cnt), cnt_stride=0 if cnt is None else cnt.stride(0),
                          D=D, d1=fwd.code_base, with_rest=int(with_rest), self_is_row=int(fwd.self_is_row),
                          self_col=_lib.ptr(fwd.self_col), lut=_lib.ptr(lut), S=_lib.ptr(S), shell=_lib.ptr(shell),
                          s_total=_lib.ptr(tot), c=_lib.ptr(c), e=_lib.ptr(e), q=_lib.ptr(q), dlut=_lib.ptr(dlut),
                          workspace=_lib.ptr(ws), workspace_bytes=ws.numel() * 8)
    _lib.check(_lib.lib().gnan_spmm_pb_pack1(pa, _lib.stream_of(dY)), "gnan_spmm_pb_pack1")
    gt = g.transposed()
    dS = pb_launch(gt, pbt, c.view(n, 1), _ones_table(dev, D), False, None, S_self=e, out_add=(q, lut[D - 1:]) if with_rest else None)
    return dS, dlut


def pb_bwd_applies(g: HopGraph, W: int, D: int):
    """The transposed graph's bucketed copy when the ONE-column backward can take the propagation-blocked route, else None."""
    if not (PB_NARROW and PB_BACKWARD and W == 1 and not g.is_dense and D <= 4 and PB_MIN_NNZ <= g.nnz <= PB_MAX_NNZ):
        return None
    pb = g.transposed().pb_plan(2)
    return pb if (pb is not None and pb.n_acc == 1) else None


def pb_bwd_launch(gt: HopGraph, pb, V: torch.Tensor, S_rows: torch.Tensor, lut: torch.Tensor, with_rest: bool,
                  rest_q: Optional[torch.Tensor] = None, rest_total: Optional[torch.Tensor] = None, add_to_rows: bool = False):
    """``gnan_spmm_pb_bwd`` over the bucketed copy ``pb = gt.pb_plan(2)`` of the transposed adjacency: ``(dS [n, 1], dlut [D])``
    from the packed rows ``V [D, n_fwd_rows, 2]`` of :func:`pack_bwd_rows` (``half = 1``, no hot rows) — the arguments of
    :func:`bwd_narrow_launch`."""
    _lib.require_device(V, S_rows, lut, pb.src)
    D = int(lut.numel())
    V = V.detach().float().contiguous().view(D, -1, 2)
    if V.shape[1] != gt.n_cols:
        raise ValueError(f"packed rows for {V.shape[1]} nodes, the transposed graph lists {gt.n_cols}")
    S_rows = Fn._rows(S_rows.detach().float())
    lut = lut.detach().float().reshape(-1).contiguous()
    dS = torch.empty((gt.n_rows, 1), dtype=torch.float32, device=V.device)
    dlut = torch.empty(D, dtype=torch.float32, device=V.device)
    block = V[pb.code_base]
    a = _lib.SpmmPbArgs(n_rows=gt.n_rows, n_cols=gt.n_cols, S=_lib.ptr(block), s_stride=2, W=2, D=D, lut=_lib.ptr(lut),
                        cnt=None, cnt_stride=0, s_total=None, Y=None, y_stride=0, n_entries=pb.n_entries, src=_lib.ptr(pb.src),
                        dst=_lib.ptr(pb.dst), cb_width=pb.cb_width, n_cblocks=pb.n_cblocks, chunk_q=_lib.ptr(pb.chunk_q),
                        cb_chunk_ptr=_lib.ptr(pb.cb_chunk_ptr), n_bins=pb.n_bins, acc_per_bin=pb.acc_per_bin,
                        bin_order=_lib.ptr(pb.bin_order), bin_entry_ptr=_lib.ptr(pb.bin_entry_ptr),
                        bin_row_ptr=_lib.ptr(pb.bin_row_ptr), slot_ptr=_lib.ptr(pb.slot_ptr), n_acc=pb.n_acc,
                        code_base=pb.code_base, self_col=_lib.ptr(pb.self_col), headroom_bits=pb.headroom_bits, flags=PB_FLAGS, self_is_row=int(pb.self_is_row))
    ga = _lib.SpmmPbBwdArgs(pb=a, v_self=_lib.ptr(V[0]) if pb.code_base else None, s_rows=_lib.ptr(S_rows),
                            s_rows_stride=S_rows.stride(0), with_rest=int(with_rest), dS=_lib.ptr(dS), ds_stride=dS.stride(0),
                            dlut=_lib.ptr(dlut))
    keep = []
    if rest_q is not None:
        rest_q = rest_q.detach().float().reshape(-1).contiguous()
        keep.append(rest_q)
        if rest_total is not None:
            rest_total = rest_total.detach().float().reshape(-1).contiguous()
            keep.append(rest_total)
            ga.rest_total, ga.rest_q = _lib.ptr(rest_total), _lib.ptr(rest_q)
        if add_to_rows:
            scale = lut[D - 1:]
            keep.append(scale)
            ga.ds_add, ga.ds_add_scale = _lib.ptr(rest_q), _lib.ptr(scale)
    need = _lib.lib().gnan_spmm_pb_bwd_workspace_bytes(ga)
    ws = torch.empty((need + 15) // 16 * 4, dtype=torch.float32, device=V.device)
    ga.pb.workspace, ga.pb.workspace_bytes = _lib.ptr(ws), ws.numel() * 4
    _lib.check(_lib.lib().gnan_spmm_pb_bwd(ga, _lib.stream_of(V)), "gnan_spmm_pb_bwd")
    return dS, dlut


def shell_sums_launch(g: HopGraph, S: torch.Tensor, lut_like: torch.Tensor, with_rest: bool,
                      row_ids: Optional[torch.Tensor] = None, s_total: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``T[q, d, w]`` = sum of operand rows per hop shell (``gnan_spmm_shell_sums``); the rest shell is ``s_total``
    (default: the column sums of ``S``) minus the listed rows."""
    S = S.detach().float()
    S = Fn._rows(S)
    n_out = g.n_rows if row_ids is None else int(row_ids.numel())
    D = g.n_codes
    T = torch.zeros((n_out, D, S.shape[1]), dtype=torch.float32, device=S.device)
    if with_rest and s_total is None:
        s_total = Fn.column_sums(S)
    if not with_rest:
        s_total = None
    lut = lut_like.detach().float().contiguous()
    a = _spmm_args(g, S, lut, False, s_total, T.view(n_out, -1), row_ids, lut.dim() == 3)
    _lib.check(_lib.lib().gnan_spmm_shell_sums(a, _lib.stream_of(S)), "gnan_spmm_shell_sums")
    return T


def lut_grad_launch(g: HopGraph, S: torch.Tensor, dY: torch.Tensor, D: int, use_cnt: bool, with_rest: bool,
                    row_ids: Optional[torch.Tensor], s_total: Optional[torch.Tensor], reduce_rows: bool) -> torch.Tensor:
    """``gnan_spmm_lut_grad``: ``dwt[q, d] = inv(q, d) * sum_w dY[q, w % dY.shape[1]] * T[q, d, w]`` without the
    ``[n, D, W]`` shell sums; ``reduce_rows`` sums over the rows -> ``[D, 1]``, else ``[n_out, D, 1]``."""
    S = S.detach().float()
    S = Fn._rows(S)
    dY = dY.detach().float().contiguous()
    n_out = g.n_rows if row_ids is None else int(row_ids.numel())
    if with_rest and s_total is None:
        s_total = Fn.column_sums(S)
    if not with_rest:
        s_total = None
    scatter = False
    if g.is_dense:
        plan = None                              # dense_lut_grad_kernel: one wave per row, no schedule, every output written
    elif row_ids is None and S.shape[1] >= DEGREE_SCHEDULE_MIN_WIDTH and g.n_rows > 1:
        if DEGREE_SORTED_COPY and not g.is_dense and g.n_rows >= DEGREE_SORTED_COPY_MIN_ROWS:
            g, row_ids, plan = g.degree_sorted_copy()   # as the forward: adjacent index ranges for neighbouring lane groups
            scatter = 2
        else:
            row_ids, plan = g.degree_schedule()
            scatter = True
    else:
        plan = g.long_row_plan(row_ids)
    out = (torch.empty if g.is_dense else torch.zeros)((D,) if reduce_rows else (n_out, D), dtype=torch.float32, device=S.device)
    lut_like = torch.empty((D, 1), dtype=torch.float32, device=S.device)       # only its shape is read
    a = _spmm_args(g, S, lut_like, use_cnt, s_total, out.view(-1, 1), row_ids, False, plan=plan, scatter_out=scatter)
    a.n_rows, a.y_stride = n_out, S.shape[1]                                   # Y is not written by this entry point
    ga = _lib.SpmmLutGradArgs(spmm=a, dY=_lib.ptr(dY), dy_stride=dY.stride(0), dy_channels=dY.shape[1],
                              reduce_rows=int(reduce_rows), dwt=_lib.ptr(out))
    need = _lib.lib().gnan_spmm_lut_grad_workspace_bytes(ga)
    ws = torch.empty(need // 8 + 1, dtype=torch.float64, device=S.device)
    ga.workspace, ga.workspace_bytes = _lib.ptr(ws), ws.numel() * 8
    _lib.check(_lib.lib().gnan_spmm_lut_grad(ga, _lib.stream_of(S)), "gnan_spmm_lut_grad")
    return out.unsqueeze(-1)


NARROW_FUSED_BACKWARD = True   # dS and dlut from one transposed pass
DENSE_LUT_GRAD = True   # dense layout: the table gradient in one pass (dense_lut_grad_kernel)
SMALL_DENSE_ROWS = 1024     # dense graphs up to this size read (lut, cnt) per pair in the operand-gradient pass: building the per-node
                            # weight table first is four more launches than the whole pass on a 30-node graph
NARROW_BWD_PERSISTENT = True   # ... one channel: packed index, persistent workgroups, hot rows in LDS


def pack_bwd_rows(dY: torch.Tensor, cnt: Optional[torch.Tensor], D: int, with_rest: bool, half: int,
                  hot: Optional[torch.Tensor] = None, want_q_sum: bool = False):
    """``V[d, i] = [dY_i / cnt(i, d) | dY_i / cnt(i, D-1)]`` (code-major ``[D, n (+ hot), 2 * half]``), halves zero padded to
    ``half`` floats (``gnan_spmm_pack_bwd_rows``); with ``hot`` (node ids) the rows of those nodes are repeated behind the n
    real ones of every code block: ``V[d, n + k] = V[d, hot[k]]``.  ``want_q_sum`` (one channel, rest bucket): returns
    ``(V, q_sum [1])`` with ``q_sum = sum_i dY_i / cnt(i, D-1)`` over the n real nodes, out of the same pass."""
    _lib.require_device(dY)
    dY = Fn._rows(dY.detach().float())
    n, W = dY.shape
    k = 0 if hot is None else int(hot.numel())
    V = torch.empty((D, n + k, 2 * half), dtype=torch.float32, device=dY.device)
    c = None if cnt is None else cnt.contiguous()
    h = None if hot is None else hot.to(torch.int64).contiguous()
    pa = _lib.PackBwdRowsArgs(dY=_lib.ptr(dY), dy_stride=dY.stride(0), W=W, D=D, cnt=_lib.ptr(c),
                              cnt_stride=0 if c is None else c.stride(0), n=n, with_rest=int(with_rest), half=half,
                              V=_lib.ptr(V), hot=_lib.ptr(h), n_hot=k)
    q_sum = None
    if want_q_sum:
        if W != 1 or not with_rest:
            raise ValueError("pack_bwd_rows: q_sum is the one-channel rest-bucket sum")
        q_sum = torch.empty(1, dtype=torch.float32, device=dY.device)
        pa.q_sum = _lib.ptr(q_sum)
        pa.q_arrive = _lib.ptr(Fn.arrive_counter(dY.device, 2))
        need = _lib.lib().gnan_spmm_pack_bwd_rows_workspace_bytes(pa)
        ws = torch.empty(max(1, need // 8), dtype=torch.float64, device=dY.device)      # (alive until the launch is queued)
        pa.q_workspace, pa.q_workspace_bytes = _lib.ptr(ws), ws.numel() * 8
    _lib.check(_lib.lib().gnan_spmm_pack_bwd_rows(pa, _lib.stream_of(dY)), "gnan_spmm_pack_bwd_rows")
    return (V, q_sum) if want_q_sum else V


def narrow_walk(g: HopGraph):
    """``(graph to walk, processing order, hot ids)`` for narrow operand rows: the degree-sorted copy of a large CSR —
    rows of equal length share a wavefront — whose column ids point the most listed neighbours at a compact copy of
    their rows behind the operand (``HopGraph.degree_sorted_copy_hot``); ``(g, None, None)`` for small or dense graphs."""
    if (NARROW_SORTED_WALK and DEGREE_SORTED_COPY and not g.is_dense and g.n_rows >= DEGREE_SORTED_COPY_MIN_ROWS
            and g.nnz >= NARROW_SORTED_MIN_NNZ):
        return g.degree_sorted_copy_hot() if HOT_COLUMN_ROWS else (*g.degree_sorted_copy()[:2], None)
    return g, None, None


def bwd_narrow_launch(gt: HopGraph, V: torch.Tensor, S_rows: torch.Tensor, lut: torch.Tensor, with_rest: bool, W: int,
                      walk=None, ds_add: Optional[torch.Tensor] = None, rest_q: Optional[torch.Tensor] = None,
                      rest_total: Optional[torch.Tensor] = None, add_to_rows: bool = False):
    """``gnan_spmm_bwd_narrow`` over the transposed adjacency ``gt``: returns ``(dS [n, W], dlut [D])`` — see the header for
    the layout of ``V [D * n_fwd_rows, 2 * half]`` (code-major).  ``walk = narrow_walk(gt)`` when the caller has already put the hot
    rows behind ``V`` (``pack_bwd_rows(hot=...)``).  The rest bucket's column-sum term ``wt(i, rest) * total``: ``rest_q [W]``
    (the column sums of the packed rows' rest halves) and ``rest_total [W]`` add ``<rest_total, rest_q>`` to ``dlut[D - 1]``;
    ``add_to_rows`` adds ``lut[D - 1] * rest_q`` to every row of ``dS`` (``ds_add``: a ready vector for the same place)."""
    _lib.require_device(V, S_rows, lut, gt.code)
    V = Fn._rows(V.detach().float())
    S_rows = Fn._rows(S_rows.detach().float())
    lut = lut.detach().float().reshape(-1, 1).contiguous()
    D = lut.shape[0]
    n_out = gt.n_rows
    dS = torch.empty((n_out, W), dtype=torch.float32, device=V.device)
    dlut = torch.empty(D, dtype=torch.float32, device=V.device)
    # rows of equal length share a wavefront, and the packed rows of the most listed nodes are read from a compact copy
    # behind V (see spmm_launch); dS is bit-identical, the table gradient adds its float64 partials in processing order
    appended = walk is not None
    gt, order, hot = walk if appended else narrow_walk(gt)
    if hot is not None and not appended:                      # V [D * n, 2 * half], code-major: the hot rows go behind every code block
        V3 = V.view(D, -1, V.shape[1])
        V = torch.cat([V3, V3.index_select(1, hot)], dim=1).view(-1, V.shape[1])
    scatter = 0 if order is None else 2
    plan = gt.narrow_row_plan() if (V.shape[1] * 4 <= 8 and NARROW_ROW_SLICING) else gt.long_row_plan()
    # one channel (packed rows of 2 floats) over a sorted copy: one packed index stream, persistent workgroups
    # (spmm_bwd_hot_kernel) — 10M-node R-MAT: 1.19 -> see DESIGN.md section 4.6
    packed = V.shape[1] == 2 and order is not None and NARROW_BWD_PERSISTENT
    a = _spmm_args(gt, V, lut, False, None, dS, order, False, plan=plan, scatter_out=scatter, packed=packed)
    a.n_cols = gt.n_cols                              # rows of V = n_cols * D (checked by the kernel's addressing only)
    a.y_stride = V.shape[1]                           # Y is not written by this entry point (dS is); keeps validate() content
    na = _lib.SpmmBwdNarrowArgs(spmm=a, s_rows=_lib.ptr(S_rows), s_rows_stride=S_rows.stride(0), w_real=W,
                                with_rest=int(with_rest), dS=_lib.ptr(dS), ds_stride=dS.stride(0), dlut=_lib.ptr(dlut),
                                ds_add=None if ds_add is None else _lib.ptr(ds_add))
    if rest_q is not None:
        rest_q = rest_q.detach().float().contiguous()
        if rest_total is not None:
            rest_total = rest_total.detach().float().contiguous()
            if rest_total.numel() != W or rest_q.numel() != W:
                raise ValueError("bwd_narrow: rest_total / rest_q hold one value per operand column")
            na.rest_total, na.rest_q = _lib.ptr(rest_total), _lib.ptr(rest_q)
        if add_to_rows:
            if ds_add is not None:
                raise ValueError("bwd_narrow: either a ready ds_add or add_to_rows")
            na.ds_add, na.ds_add_scale = _lib.ptr(rest_q), _lib.ptr(lut[D - 1:])
    if a.packed_index and hot is not None and HOT_ROWS_IN_LDS:
        # ... with the head of the appended hot rows in LDS.  Code 0 is the self pair of a hop-coded graph (one pair per
        # row, never a hot one): the LDS copy covers the other listed codes
        listed = D - 1 if with_rest else D
        code_lo = 1 if listed > 1 else 0
        codes = listed - code_lo
        head = HOT_LDS_FLOATS // (2 * codes)
        head = 1 << (head.bit_length() - 1)           # the shares are known for 4096 / 8192 / 16384 rows
        head = min(int(hot.numel()), head)
        share = (getattr(gt, "_hot_head_share", None) or {}).get(head, 0.0)
        if share >= HOT_LDS_MIN_SHARE:
            a.hot_lo, a.hot_rows = gt.n_cols - int(hot.numel()), head
            na.spmm.hot_lo, na.spmm.hot_rows = a.hot_lo, a.hot_rows       # (the struct was copied into na)
            na.hot_code_lo, na.hot_codes = code_lo, codes
    need = _lib.lib().gnan_spmm_bwd_narrow_workspace_bytes(na)
    ws = torch.empty(need // 8 + 2, dtype=torch.float64, device=V.device)
    na.workspace, na.workspace_bytes = _lib.ptr(ws), ws.numel() * 8
    _lib.check(_lib.lib().gnan_spmm_bwd_narrow(na, _lib.stream_of(V)), "gnan_spmm_bwd_narrow")
    return dS, dlut


class _NotShared:
    """Sentinel: the rest-bucket total is this process's own (``None`` already means "the default process group")."""
    def __repr__(self):
        return "NOT_SHARED"


NOT_SHARED = _NotShared()


class _RhoAggregate(torch.autograd.Function):
    """Y = A_w(lut, cnt) @ S  with the rest-bucket term; gradients for S and the weight table."""

    @staticmethod
    def forward(ctx, S, lut, g: HopGraph, use_cnt: bool, with_rest: bool, row_ids, s_total=None, reduce_cr=0,
                total_rows=None, total_group=NOT_SHARED):
        ctx.g, ctx.use_cnt, ctx.with_rest, ctx.row_ids, ctx.reduce_cr = g, use_cnt, with_rest, row_ids, reduce_cr
        ctx.s_total = None if s_total is None else s_total.detach()
        ctx.total_rows, ctx.total_group = total_rows, total_group
        ctx.save_for_backward(S, lut)
        kept = [] if (S.shape[1] == 1 and all(ctx.needs_input_grad[:2]) and S.dtype == torch.float32) else None
        out = spmm_launch(g, S, lut, use_cnt, with_rest, row_ids, s_total=s_total, reduce_cr=reduce_cr,
                          room=getattr(S, "gnan_room", None), keep_shell=kept)
        ctx.shell = kept[0] if kept else None
        return out

    @staticmethod
    def backward(ctx, dY):
        S, lut = ctx.saved_tensors
        dS, dlut = _aggregate_backward(ctx, S, lut, dY, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dS, dlut, None, None, None, None, None, None, None, None


def _aggregate_backward(ctx, S, lut, dY, need_dS: bool, need_dlut: bool):
    """Gradients of ``Y = A_w(lut, cnt) @ S`` (+ rest bucket) w.r.t. the operand and the weight table; ``ctx`` carries
    ``g, use_cnt, with_rest, row_ids, reduce_cr, s_total, total_rows, total_group`` as :class:`_RhoAggregate` stores them."""
    g, use_cnt, with_rest, row_ids = ctx.g, ctx.use_cnt, ctx.with_rest, ctx.row_ids
    dY = dY.contiguous().float()
    dY_out = dY                           # as the forward returned it: [n_out, W] or, with the fused sum, [n_out, cr]
    W = S.shape[1]
    D, Cw = lut.shape[-2], lut.shape[-1]
    per_row = lut.dim() == 3
    # truncated-hop graphs: the table gradient comes out of one pass over the listed pairs (gnan_spmm_lut_grad)
    fused_lut_grad = (need_dlut and not g.is_dense and D <= 4 and Cw == 1
                      and S.dtype == torch.float32)
    # dense layout (every pair listed, up to 256 shells), global table: one pass as well (dense_lut_grad_kernel) — the
    # shell-sum route below goes through a [n, D, W] tensor and six framework launches
    dense_lut_grad = (need_dlut and DENSE_LUT_GRAD and g.is_dense and Cw == 1 and D <= 256 and not per_row and not with_rest
                      and S.dtype == torch.float32)
    if ctx.reduce_cr and (need_dS or (need_dlut and not (fused_lut_grad or dense_lut_grad))):
        dY = dY.repeat(1, W // ctx.reduce_cr)   # the fused feature sum broadcasts its gradient over the features
    rows = None if row_ids is None else row_ids.long()
    _inv = []

    def inv_counts():                       # [n_out, D] 1 / shell size — three element-wise passes over N x D: only where needed
        if not _inv:
            cnt = g.cnt if rows is None else g.cnt[rows]
            _inv.append(1.0 / cnt.clamp_min(1).float())
        return _inv[0]
    dS = dlut = None
    rest_added = False
    fused_bwd = (NARROW_FUSED_BACKWARD and need_dS and need_dlut and not g.is_dense
                 and Cw == 1 and D <= 4 and not per_row and rows is None and not ctx.reduce_cr and W <= 16
                 and S.dtype == torch.float32)
    if fused_bwd:
        # narrow operand, both gradients wanted: ONE pass over the transposed adjacency gathers, per pair, the packed row
        # [dY_i / cnt(i, d) | dY_i / cnt(i, rest)] and yields the operand gradient AND the table gradient (the two-pass
        # route below traverses the same pairs twice: 1.89 + 2.01 ms on the 10M-node graph)
        half = 1 << max(0, (W - 1).bit_length())
        shares_total = with_rest and not (ctx.total_group is NOT_SHARED and ctx.total_rows is None)
        kept = getattr(ctx, "shell", None)
        if kept is not None and kept.dim() == 2 and ROWS_BACKWARD_ONE_COLUMN and W == 1 and not shares_total:
            # the row-parallel forward kept its per-code shell sums: a pass over the rows and one gather over the transposed pairs
            total = None
            if with_rest:
                total = (ctx.s_total if ctx.s_total is not None else Fn.column_sums(S)).float().reshape(-1).contiguous()
            dS, dl = rows_bwd1_launch(g, dY, kept, lut[:, 0], use_cnt, with_rest, total)
            return dS, dl.view(D, 1)
        plans = pb_bwd1_applies(g, W, D, with_rest, not shares_total, kept if (kept is not None and kept.dim() == 1) else None)
        if plans is not None:
            # one column, large graph, shell sums kept by the forward: a pass over the rows and ONE column through the buckets
            total = None
            if with_rest:
                total = (ctx.s_total if ctx.s_total is not None else Fn.column_sums(S)).float().reshape(-1).contiguous()
            dS, dl = pb_bwd1_launch(g, plans, dY, S, ctx.shell, lut[:, 0], use_cnt, with_rest, total)
            rest_added = with_rest
            return dS, dl.view(D, 1)
        pb_t = pb_bwd_applies(g, W, D)                      # one column, large graph: no per-pair gather (csrc/spmm_pb.hip)
        walk = (None, None, None) if pb_t is not None else narrow_walk(g.transposed())
        q_sum = total = None
        if with_rest and W == 1:                            # ... and their rest halves' column sum out of the same pass
            V, q_sum = pack_bwd_rows(dY, g.cnt if use_cnt else None, D, with_rest, half, hot=walk[2], want_q_sum=True)
        else:
            V = pack_bwd_rows(dY, g.cnt if use_cnt else None, D, with_rest, half, hot=walk[2])  # [D, n (+ hot), 2 * half]
        add_to_rows = False
        if with_rest:
            if q_sum is None:
                q_sum = Fn.column_sums(V[0, :g.n_rows, half:half + W])                        # sum_i dY_i / cnt(i, rest)
            # d/dS_j of  wt(i, rest) * total : the same vector rho(0) * q_sum for every j — added by the kernel's epilogue
            add_to_rows = rest_added = ctx.total_group is NOT_SHARED and ctx.total_rows is None
            # d/d lut[rest] of the same term: <total, q_sum> — added by the kernel's final pass
            total = (ctx.s_total if ctx.s_total is not None else Fn.column_sums(S)).float().reshape(-1).contiguous()
        if pb_t is not None:
            dS, dl = pb_bwd_launch(g.transposed(), pb_t, V, S, lut[:, 0], with_rest, rest_q=q_sum, rest_total=total,
                                   add_to_rows=add_to_rows)
        else:
            dS, dl = bwd_narrow_launch(g.transposed(), V.view(-1, 2 * half), S, lut[:, 0], with_rest, W, walk=walk,
                                       rest_q=q_sum, rest_total=total, add_to_rows=add_to_rows)
        dlut = dl.view(D, 1)

    if need_dS and not fused_bwd:
        dY_full = dY
        if rows is not None:
            dY_full = torch.zeros((g.n_rows, W), dtype=torch.float32, device=dY.device)
            dY_full.index_add_(0, rows, dY)
        if not g.is_dense and Cw == 1 and W * D <= NARROW_DS_MAX_WIDTH:
            # narrow operand: fold the per-pair weight into a pre-weighted operand with one row per (node, hop code),
            # Z[i, d] = (wt(i, d) - wt(i, rest)) dY[i], and gather it over the transposed adjacency with unit weights —
            # one random request per listed pair instead of the operand row plus the neighbour's table row
            wt = (lut[..., 0] if per_row else lut[:, 0].unsqueeze(0)).float()                 # [N or 1, D]
            if use_cnt:
                wt = wt / g.cnt.clamp_min(1).float()
            if with_rest:
                wt = wt - wt[:, D - 1:D]
            Z = (wt.unsqueeze(-1) * dY_full.unsqueeze(1)).reshape(g.n_rows * D, W)
            dS = spmm_launch(g.transposed(), Z, torch.ones((D, 1), device=Z.device), False, False, None,
                             s_by_code=True)
        elif g.n_rows * D * Cw * 4 <= WEIGHT_TABLE_MAX_BYTES and not (g.is_dense and g.n_rows <= SMALL_DENSE_ROWS and not per_row):
            # wide operand: the weight of a pair belongs to the NEIGHBOUR's row there.  Read from (lut, cnt) that is two
            # random count reads, two divisions and a subtraction per pair; a per-node table wt(i, d) - wt(i, rest)
            # built once per backward pass makes it one 4-byte read (arxiv-shaped W = 40: 0.55 -> 0.28 ms)
            if not per_row and lut.is_cuda and g.cnt.dtype == torch.int32:
                wt = Fn.weight_table(lut, g.cnt if use_cnt else None, g.n_rows, with_rest)    # one launch (four framework ones before)
            else:
                wt = (lut if per_row else lut.unsqueeze(0)).float()                           # [N or 1, D, Cw]
                if use_cnt:
                    wt = wt / g.cnt.clamp_min(1).float().unsqueeze(-1)
                if with_rest:
                    wt = wt - wt[:, D - 1:D]
                wt = wt.expand(g.n_rows, D, Cw).contiguous()
            dS = spmm_launch(g.transposed(), dY_full, wt, False, False, None, weight_by_col=True)
        else:
            dS = spmm_launch(g.transposed(), dY_full, lut, use_cnt, False, None,
                             weight_by_col=True, minus_rest=with_rest)
    if need_dS:
        if with_rest and not rest_added:
            # d/dS_j of  wt(i, rest) * total  : the same vector for every j
            if (not per_row and rows is None and Cw == 1 and use_cnt and dY.is_cuda and g.cnt.dtype == torch.int32
                    and dY.shape[0] == g.n_rows):
                # rho(0) sum_i dY[i, :] / cnt(i, rest) in one weighted column sum (eight framework launches before)
                v = Fn.column_sums_weighted(dY, g.cnt[:, D - 1], lut[D - 1]).view(1, W)
            else:
                l_rest = (lut[rows, D - 1] if rows is not None else lut[:, D - 1]) if per_row else lut[D - 1].unsqueeze(0)
                w_rest = l_rest * inv_counts()[:, D - 1:D] if use_cnt else l_rest   # [n_out or 1, Cw]
                w_rest = w_rest.expand(dY.shape[0], Cw).repeat(1, W // Cw)
                v = (w_rest * dY).sum(0, keepdim=True)
            if ctx.total_group is not NOT_SHARED:
                # the total was summed over the ranks of a group: every rank's output rows pull on every rank's
                # summed operand rows, so the ranks add their vectors (W floats) before handing them down
                import torch.distributed as dist
                dist.all_reduce(v, op=dist.ReduceOp.SUM, group=ctx.total_group)
            if ctx.total_rows is None:
                dS.add_(v)
            else:                          # only the first rows of S went into the total (owned rows ahead of halo rows)
                dS[: ctx.total_rows] += v

    if fused_bwd:
        pass                                  # both gradients came out of the one transposed pass above
    elif need_dlut and (dense_lut_grad or (fused_lut_grad and not per_row)):
        dlut = lut_grad_launch(g, S, dY_out, D, use_cnt, with_rest, row_ids, ctx.s_total, True)       # [D, 1]
    elif need_dlut:
        if fused_lut_grad:
            dwt = lut_grad_launch(g, S, dY_out, D, use_cnt, with_rest, row_ids, ctx.s_total, False)   # [n_out, D, 1]
        else:
            T = shell_sums_launch(g, S, lut, with_rest, row_ids, ctx.s_total)  # [n_out, D, W]
            dwt = (T.view(T.shape[0], D, W // Cw, Cw) * dY.view(dY.shape[0], 1, W // Cw, Cw)).sum(2)
            if use_cnt:
                dwt = dwt * inv_counts().unsqueeze(-1)                        # [n_out, D, Cw]
        if per_row:
            if rows is None:
                dlut = dwt
            else:
                dlut = torch.zeros_like(lut)
                dlut.index_add_(0, rows, dwt)
        else:
            dlut = dwt.sum(0)
    return dS, dlut


class _Bag:
    pass


class _PreRhoAggregate(torch.autograd.Function):
    """``Y[i] = sum_j rho(u_ij / c_ij) (.) S[j]`` — GNAN.py:64-70 with the pre-rho normalisation of GNAN.py:65-67 — from
    rho's table: the per-row weights are looked up (``gnan_rho_row_lut``) for the rows in the order the aggregation walks
    them, nothing is permuted and no table of the natural order exists in the forward pass.  Backward: the natural-order
    table and its arguments (one more launch), the aggregation's own gradients, then rho's parameter gradients from the
    gradient of the table binned by the pieces of its arguments."""

    @staticmethod
    def forward(ctx, S, g, with_rest, row_ids, s_total, total_rows, total_group, tables, u, L, H, C, *params):
        ctx.g, ctx.use_cnt, ctx.with_rest, ctx.row_ids, ctx.reduce_cr = g, False, with_rest, row_ids, 0
        ctx.s_total = None if s_total is None else s_total.detach()
        ctx.total_rows, ctx.total_group = total_rows, total_group
        ctx.tables, ctx.u, ctx.meta = tables, u, (L, H, C)
        ctx.present = [t is not None for t in params]
        ctx.save_for_backward(S, *[t for t in params if t is not None])
        return spmm_launch(g, S, None, False, with_rest, row_ids, s_total=s_total, room=getattr(S, "gnan_room", None),
                           lut_of_counts=lambda cnt: Fn._rho_row_lut_launch(cnt, u, tables, C, False)[0], lut_channels=C)

    @staticmethod
    def backward(ctx, dY):
        L, H, C = ctx.meta
        saved = list(ctx.saved_tensors)
        S = saved.pop(0)
        params = [saved.pop(0) if present else None for present in ctx.present]
        need_rho = any(ctx.needs_input_grad[12:])
        lut, arg = Fn._rho_row_lut_launch(ctx.g.cnt, ctx.u, ctx.tables, C, need_rho)
        dS, dlut = _aggregate_backward(ctx, S, lut, dY, ctx.needs_input_grad[0], need_rho)
        pg = Fn._rho_param_grads(arg, dlut, ctx.tables, params, ctx.present, L, H, C) if need_rho else (None,) * 6
        return (dS,) + (None,) * 11 + tuple(pg)


def pre_rho_aggregate(g: HopGraph, S: torch.Tensor, p: StackedMLP, u: torch.Tensor, with_rest: Optional[bool] = None,
                      row_ids: Optional[torch.Tensor] = None, s_total: Optional[torch.Tensor] = None,
                      total_rows: Optional[int] = None, total_group=NOT_SHARED) -> torch.Tensor:
    """The aggregation with the pre-rho normalisation of the stand-alone model file (GNAN.py:65-70):
    ``Y[q] = sum_j rho(u(i_q, j) / c(i_q, j)) (.) S[j]``; ``p`` = rho's layers as a one-feature :class:`StackedMLP`,
    ``u`` = the distinct values of ``node_distances`` (``graph.hop_inputs``).  Differentiable w.r.t. ``S`` and rho."""
    if with_rest is None:
        with_rest = not g.is_dense
    if row_ids is not None:
        row_ids = row_ids.to(device=g.device, dtype=torch.int32).contiguous()
    tables = Fn._rho_tables(p, g.n_rows * g.n_codes) if S.dtype == torch.float32 else None
    if tables is None:                      # small graphs / graph capture: the table through the shape-function kernels
        return rho_aggregate(g, S, Fn.rho_row_lut(g.cnt, u, p), False, with_rest, row_ids, s_total,
                             total_rows=total_rows, total_group=total_group)
    return _PreRhoAggregate.apply(S, g, with_rest, row_ids, s_total, total_rows, total_group, tables, u.contiguous(),
                                  p.L, p.H, p.C, *p[:6])


REFERENCE_ORDER_KEEP_MAX_BYTES = 16 << 30   # the [N, F*C] rows of a reference-order forward are kept for its backward below this


class _ReferenceOrderAggregate(torch.autograd.Function):
    """The reference's evaluation order with the feature sum fused (models.py:360-376): ``fx = f(x)`` per feature,
    ``Y[i, c] = sum_k sum_j m_ij fx[j, k, c]``, as ONE autograd node over the table path.

    Forward: look-up of the ``[N, F*C]`` rows, aggregation with the read-out in its epilogue (as :func:`feature_mlps` +
    :func:`rho_aggregate`).  Backward: ``Y`` depends on the rows only through their feature sum ``S1 = sum_k fx[:, k, :]``, so
    ``d loss / d fx[j, k, c] = (A^T dY)[j, c]`` for EVERY feature k and the table gradient is that of the narrow aggregation
    of ``S1`` — the backward pass of the sum-first order: one narrow transposed pass for both aggregation gradients, and the
    per-piece moments of all features from the one ``[N, C]`` gradient.  The composed nodes walk the transposed graph with
    ``[N, F*C]`` rows, contract a second wide pass for the table, and bin an ``[N, F*C]`` gradient whose F blocks are
    equal (10M-node graph, F = 64: 20 ms of backward against 3)."""

    @staticmethod
    def forward(ctx, x, lut, g, use_cnt, with_rest, L, H, C, F, *params):
        p = StackedMLP(*params, L, H, C, F)
        needs_grad = any(ctx.needs_input_grad[9:])
        located = None
        fx, tables, total = Fn._fmlp_forward(x, p, False, with_rest, needs_grad, torch.float32, None, located=located)
        if with_rest and total is None:
            total = Fn.column_sums(fx)
        ctx.g, ctx.use_cnt, ctx.with_rest, ctx.row_ids, ctx.reduce_cr = g, use_cnt, with_rest, None, 0
        ctx.total_rows, ctx.total_group = None, NOT_SHARED
        ctx.tables, ctx.meta = tables, (L, H, C, F)
        ctx.present = [t is not None for t in params]
        keep = fx if (fx.numel() * 4 <= REFERENCE_ORDER_KEEP_MAX_BYTES or tables is None) else None
        ctx.kept_rows = keep is not None
        ctx.x_abs_max = Fn._abs_max_cached(x) if needs_grad and x.numel() else None
        ctx.save_for_backward(x, lut, *([keep] if keep is not None else []), *[t for t in params if t is not None])
        return spmm_launch(g, fx, lut, use_cnt, with_rest, None, s_total=total if with_rest else None, reduce_cr=C)

    @staticmethod
    def backward(ctx, dY):
        L, H, C, F = ctx.meta
        saved = list(ctx.saved_tensors)
        x, lut = saved.pop(0), saved.pop(0)
        fx = saved.pop(0) if ctx.kept_rows else None
        params = [saved.pop(0) if present else None for present in ctx.present]
        n = x.shape[0]
        if fx is not None and fx.shape[1] % 4 == 0 and fx.stride(0) % 4 == 0 and fx.data_ptr() % 16 == 0:
            # [N, C] feature sum of the kept rows (padded features are zero columns): one streaming pass over the 2.56 GB
            S1 = Fn.feature_sum(fx, C)
        else:                                                       # rows too large to keep: the feature sum is looked up again
            S1 = Fn._fpwl_launch(x, ctx.tables, True)
        ctx.s_total = None
        dS1, dlut = _aggregate_backward(ctx, S1, lut, dY, True, ctx.needs_input_grad[1])
        pg = (None,) * 6
        if any(ctx.needs_input_grad[9:]):
            # every feature's rows have the gradient dS1: the shape functions' gradients are those of the feature-SUM mode
            _, pg = Fn._shape_function_grads(x, params, ctx.present, ctx.tables, dS1, True, L, H, C, F, ctx.x_abs_max)
        return (None, dlut, None, None, None, None, None, None, None, *pg)


def reference_order_applies(x: torch.Tensor, p: StackedMLP, lut: torch.Tensor, g: HopGraph) -> bool:
    """Can :func:`reference_order_forward` take this call?  Training through the table path (what AUTO picks from 2^18
    look-ups), a global weight table, a read-out width the aggregation kernel fuses, features without a gradient."""
    grads = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in p[:6])
    return (grads and not x.requires_grad and lut.dim() == 2 and p.C in FUSABLE_READOUT and p.L >= 2 and not g.is_dense
            and (Fn.FMLP_ALGO == _lib.FMLP_PWL or (Fn.FMLP_ALGO == _lib.FMLP_AUTO and x.shape[0] * p.F >= Fn.PWL_MIN_WORK_GRAD))
            and not torch.cuda.is_current_stream_capturing() and not (Fn.PAD_FEATURES and p.F % Fn.PAD_FEATURES and p.C == 1
                                                                      and x.shape[0] * p.F >= Fn.PAD_MIN_WORK))


def reference_order_forward(g: HopGraph, x: torch.Tensor, p: StackedMLP, lut: torch.Tensor, use_cnt: bool) -> torch.Tensor:
    """``Y [N, C]`` in the reference's evaluation order (per-feature rows, aggregate, sum over features) as one autograd node
    whose backward pass is the sum-first order's (see :class:`_ReferenceOrderAggregate`).  Check :func:`reference_order_applies`."""
    _lib.require_device(x, p.w_last, lut)
    return _ReferenceOrderAggregate.apply(x, lut, g, use_cnt, not g.is_dense, p.L, p.H, p.C, p.F, *p[:6])


def add_rest_total_term(Y: torch.Tensor, g: HopGraph, lut: torch.Tensor, use_cnt: bool, total: torch.Tensor,
                        reduce_channels: int = 0, row_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``Y[q, c] += wt(i_q, D-1, .) * total`` in place (``gnan_rest_term_add``): the part of the aggregation that depends on
    the operand only through its column sums.

    ``rho_aggregate(..., s_total=total) == add_rest_total_term(rho_aggregate(..., s_total=zeros), ..., total)`` up to
    rounding: a multi-rank forward runs the aggregation while the all-reduce of ``total`` is still in flight and adds this
    term afterwards (inference only: no autograd through it)."""
    _lib.require_device(Y, lut, total)
    D, Cw = lut.shape[-2], lut.shape[-1]
    W = int(total.numel())
    lutc, tot = Fn._c(lut.detach().float()), Fn._c(total.detach().float())
    cnt = g.cnt if use_cnt else None
    rows = None if row_ids is None else row_ids.to(device=Y.device, dtype=torch.int32).contiguous()
    a = _lib.RestTermArgs(Y=_lib.ptr(Y), y_stride=Y.stride(0), n=Y.shape[0], total=_lib.ptr(tot), W=W, lut=_lib.ptr(lutc),
                          lut_row_stride=(D * Cw if lut.dim() == 3 else 0), D=D, Cw=Cw, cnt=_lib.ptr(cnt),
                          cnt_stride=0 if cnt is None else cnt.stride(0), row_ids=_lib.ptr(rows), reduce_cr=int(reduce_channels))
    _lib.check(_lib.lib().gnan_rest_term_add(a, _lib.stream_of(Y)), "gnan_rest_term_add")
    return Y


def rest_total_term(g: HopGraph, lut: torch.Tensor, use_cnt: bool, total: torch.Tensor, reduce_channels: int = 0,
                    row_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The term of :func:`add_rest_total_term` on its own: ``R[q, c]`` (``[n, reduce_channels or W]``)."""
    n = g.n_rows if row_ids is None else int(row_ids.numel())
    R = torch.zeros((n, reduce_channels or int(total.numel())), dtype=torch.float32, device=total.device)
    return add_rest_total_term(R, g, lut, use_cnt, total, reduce_channels, row_ids)


def rho_aggregate(g: HopGraph, S: torch.Tensor, lut: torch.Tensor, use_cnt: bool,
                  with_rest: Optional[bool] = None, row_ids: Optional[torch.Tensor] = None,
                  s_total: Optional[torch.Tensor] = None, reduce_channels: int = 0,
                  total_rows: Optional[int] = None, total_group=NOT_SHARED) -> torch.Tensor:
    """``Y[q] = sum_j wt(i_q, hop(i_q, j)) * S[j]`` over the hop-coded adjacency ``g``.

    ``lut [D, Cw]`` (post-rho / un-normalised: ``rho`` at the D distinct distances) or
    ``lut [N, D, Cw]`` (pre-rho: ``rho(u_d / cnt[i, d])``); ``use_cnt`` divides by the shell size
    (models.py:369-370).  ``with_rest`` defaults to True for CSR graphs (unlisted pairs get the
    ``rho(0)`` weight) and False for dense ones (every pair is listed).

    ``s_total`` replaces the column sums of ``S`` as the rest bucket's total; it is treated as a constant by autograd and
    its dependence on ``S`` is accounted for in this operator's backward: ``total_rows`` says that only the first rows of
    ``S`` were summed into it (a rank's owned rows ahead of its halo rows), ``total_group`` that it was then summed over
    the ranks of that process group (``None`` = the default group) — the backward pass all-reduces the matching W-float
    vector over the same group, so EVERY rank of the group must run it.
    """
    if with_rest is None:
        with_rest = not g.is_dense
    if S.dtype == torch.bfloat16 and (S.requires_grad or (torch.is_grad_enabled() and lut.requires_grad)):
        raise _lib.GnanHipError("bf16 operand storage is an inference format: run under torch.no_grad()")
    if row_ids is not None:
        row_ids = row_ids.to(device=g.device, dtype=torch.int32).contiguous()
    if reduce_channels and reduce_channels not in FUSABLE_READOUT:
        Y = _RhoAggregate.apply(S, lut, g, use_cnt, with_rest, row_ids, s_total, 0, total_rows, total_group)
        return Y.view(Y.shape[0], -1, reduce_channels).sum(dim=1)
    return _RhoAggregate.apply(S, lut, g, use_cnt, with_rest, row_ids, s_total, reduce_channels, total_rows, total_group)
