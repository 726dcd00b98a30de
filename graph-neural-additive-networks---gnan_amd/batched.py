"""Mirror of the reference's batched variant: ``from gnan_amd.batched import TensorGNAN`` (batched_pyg_main.py:98-184).

Many graphs per call: ``dist_batch`` is the block-diagonal matrix of raw hop counts the reference's collate
function builds (batched_pyg_main.py:54-91), ``-1`` marking cross-graph / unreachable pairs.  rho acts on the raw
hop count, there is no shell normalisation, masked pairs contribute nothing (batched_pyg_main.py:151-159), and node
outputs are summed per graph (batched_pyg_main.py:173-181).

The path here never needs the ``(sum N_g)^2`` matrix:

* :func:`collate` is the device-side counterpart of ``distance_collate_fn``: the per-graph hop matrices become
  :class:`HopBlocks` — packed ``[n_g, n_g]`` uint8 code blocks + offsets (``gnan_hops_to_code``); a dense ``dist_batch``
  handed to ``forward`` as the reference does is cut into the same blocks by one kernel (``gnan_dense_blocks_to_code``,
  which also checks that nothing outside the diagonal blocks is listed) and remembered per tensor;
* the forward of ALL graphs of a small batch (up to ~1500 nodes: 32-48 Mutagenicity-sized graphs) is ONE launch
  (``gnan_small_batch_fwd``: per graph the workgroups of ``gnan_small_graph_fwd`` — shape functions, rho on the distinct hop
  counts, aggregation — and the per-graph read-out in the epilogue of the graph's last workgroup); larger batches, graphs of
  more than 128 nodes, or training-mode Dropout take the hop-coded CSR of the blocks through the general kernels (flat
  ~0.13 ms per batch up to 512 graphs) and ``gnan_segment_sum`` for the per-graph read-out: no ``scatter_add`` either way;
* backward: the general kernels on the CSR (transposed aggregation, table gradient, ``gnan_fmlp_bwd`` twice) from the
  node sums and the rho table the forward left behind.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import nn

from . import _lib
from .aggregate import rho_aggregate
from .graph import HopGraph
from .modules import _PathBase
from .small_graph import _small_mlp

BATCH_KERNEL = True            # the one-launch forward where it applies (tests compare with the CSR route)
BATCH_KERNEL_MAX_NODES = 128
# ... and pays: the launch holds F + 1 workgroups per GRAPH (built for latency: 0.05 ms for one graph, 0.09 ms for 32), the
# general kernels over the blocks' CSR take ~0.13 ms whatever the batch (tools/batched_bench.py: 128 graphs / 3.8k nodes:
# 0.28 against 0.14 ms; 512 graphs: 1.12 against 0.14 ms = 3.6M graphs/s)
BATCH_KERNEL_MAX_TOTAL_NODES = 1536


class _SegmentSum(torch.autograd.Function):
    """``out[g, :] = sum of Y's rows of graph g`` (``gnan_segment_sum``: a wave per graph, fixed order) — the scatter_add_ of
    batched_pyg_main.py:173-181; backward: every node receives its graph's gradient (an index_select, no arithmetic)."""

    @staticmethod
    def forward(ctx, Y, blocks):
        Yc = Y.detach().float()
        Yc = Yc if Yc.stride(1) == 1 else Yc.contiguous()
        out = torch.empty((blocks.n_graphs, Yc.shape[1]), dtype=torch.float32, device=Yc.device)
        _lib.check(_lib.lib().gnan_segment_sum(_lib.ptr(Yc), Yc.stride(0), Yc.shape[1], _lib.ptr(blocks.node_off),
                                               blocks.n_graphs, _lib.ptr(out), _lib.stream_of(Yc)), "gnan_segment_sum")
        ctx.blocks = blocks
        return out.to(Y.dtype)

    @staticmethod
    def backward(ctx, d):
        return d.index_select(0, ctx.blocks.batch_vector()), None


def _two_layer(hidden: int, out: int, bias: bool, dropout: float) -> nn.Sequential:
    """Linear(1,H) / ReLU / Dropout / Linear(H,out): keys 0 and 3 (batched_pyg_main.py:117-131)."""
    return nn.Sequential(nn.Linear(1, hidden, bias=bias), nn.ReLU(), nn.Dropout(dropout),
                         nn.Linear(hidden, out, bias=bias))


class HopBlocks:
    """The hop matrices of a batch of graphs on the device: ``code`` (uint8, the ``[n_g, n_g]`` blocks back to back; 255 =
    not listed), ``node_off`` int32 ``[G + 1]``, ``code_off`` int64 ``[G + 1]`` — what batched_pyg_main.py:69-76 spreads
    over a ``[sum n_g, sum n_g]`` float matrix."""

    def __init__(self, code, node_off, code_off, sizes, max_hop):
        self.code, self.node_off, self.code_off = code, node_off, code_off
        self.sizes = [int(s) for s in sizes]
        self.n_graphs = len(self.sizes)
        self.total_nodes = sum(self.sizes)
        self.max_nodes = max(self.sizes) if self.sizes else 0
        self.min_nodes = min(self.sizes) if self.sizes else 0      # 0: a gap in the graph ids (an empty graph; the reference's scatter_add gives it zeros)
        self.n_codes = int(max_hop) + 2                      # hop counts 0 .. max_hop + the (weightless) rest code
        self._csr = None
        self._batch_vector = None

    @staticmethod
    def _offsets(sizes, device):
        n = torch.tensor([0] + list(sizes), dtype=torch.int64)
        node_off = torch.cumsum(n, 0)
        code_off = torch.cumsum(n * n, 0)
        return node_off.to(torch.int32).to(device), code_off.to(device)

    @staticmethod
    def _check(status, what):
        flags, max_hop = (int(v) for v in status.tolist())
        if flags & 1:
            raise _lib.GnanHipError(f"{what} must hold integer hop counts in [0, 254] or -1")
        if flags & 2:
            raise _lib.GnanHipError(f"{what} lists a pair of nodes of two different graphs: not block-diagonal")
        return max_hop

    @staticmethod
    def from_blocks(dists: Sequence[torch.Tensor]) -> "HopBlocks":
        """From the per-graph ``[n_g, n_g]`` hop matrices (device tensors; batched_pyg_main.py:19-48)."""
        _lib.require_device(*dists)
        sizes = [int(d.shape[0]) for d in dists]
        dev = dists[0].device
        flat = torch.cat([d.reshape(-1).float() for d in dists]) if dists else torch.empty(0, device=dev)
        code = torch.empty(flat.numel(), dtype=torch.uint8, device=dev)
        status = torch.zeros(2, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().gnan_hops_to_code(_lib.ptr(flat), flat.numel(), _lib.ptr(code), _lib.ptr(status),
                                                _lib.stream_of(code)), "gnan_hops_to_code")
        node_off, code_off = HopBlocks._offsets(sizes, dev)
        return HopBlocks(code, node_off, code_off, sizes, HopBlocks._check(status, "the hop matrices"))

    @staticmethod
    def from_dense(dist_batch: torch.Tensor, batch_vector: torch.Tensor) -> "HopBlocks":
        """From the reference's dense block-diagonal ``dist_batch`` (batched_pyg_main.py:69-76) and ``batch_vector``
        (sorted: node -> graph, batched_pyg_main.py:84-88)."""
        _lib.require_device(dist_batch)
        if dist_batch.dim() != 2 or dist_batch.shape[0] != dist_batch.shape[1]:
            raise ValueError(f"dist_batch must be square, got {tuple(dist_batch.shape)}")
        dev = dist_batch.device
        bv = batch_vector.to(dev).long()
        if bv.numel() != dist_batch.shape[0]:
            raise ValueError("batch_vector must name the graph of every row of dist_batch")
        sizes = torch.bincount(bv).tolist() if bv.numel() else []
        if bv.numel() and bool((bv[1:] < bv[:-1]).any()):
            raise _lib.GnanHipError("batch_vector must be sorted (the nodes of a graph are consecutive rows)")
        node_off, code_off = HopBlocks._offsets(sizes, dev)
        d = dist_batch.float()
        d = d if d.stride(1) == 1 else d.contiguous()
        code = torch.empty(sum(s * s for s in sizes), dtype=torch.uint8, device=dev)
        status = torch.zeros(2, dtype=torch.int32, device=dev)
        graph_of = bv.to(torch.int32)
        _lib.check(_lib.lib().gnan_dense_blocks_to_code(_lib.ptr(d), d.stride(0), d.shape[0], _lib.ptr(graph_of),
                                                        _lib.ptr(node_off), _lib.ptr(code_off), _lib.ptr(code), _lib.ptr(status),
                                                        _lib.stream_of(code)), "gnan_dense_blocks_to_code")
        blocks = HopBlocks(code, node_off, code_off, sizes, HopBlocks._check(status, "dist_batch"))
        blocks._batch_vector = bv
        return blocks

    def batch_vector(self) -> torch.Tensor:
        if getattr(self, "slots", False):
            raise _lib.GnanHipError("these blocks are the slots of a captured step: their sizes live on the device only")
        if self._batch_vector is None:
            self._batch_vector = torch.repeat_interleave(torch.arange(self.n_graphs, device=self.code.device),
                                                         torch.tensor(self.sizes, device=self.code.device))
        return self._batch_vector

    def csr(self) -> HopGraph:
        """The hop-coded CSR of the listed pairs (global node ids) — what the general kernels walk: index work on the
        ``sum n_g^2`` packed codes, never on ``(sum n_g)^2`` pairs."""
        if getattr(self, "slots", False):
            raise _lib.GnanHipError("these blocks are the slots of a captured step: their sizes live on the device only")
        if self._csr is None:
            dev = self.code.device
            sizes = torch.tensor(self.sizes, dtype=torch.int64, device=dev)
            node_off = self.node_off.long()
            g_of_entry = torch.repeat_interleave(torch.arange(self.n_graphs, device=dev), sizes * sizes)
            local = torch.arange(self.code.numel(), device=dev) - self.code_off[g_of_entry]
            n_g = sizes[g_of_entry]
            row = node_off[g_of_entry] + local // n_g
            col = node_off[g_of_entry] + local % n_g
            listed = self.code != 255
            rows, cols, codes = row[listed], col[listed], self.code[listed]        # entry order == (row, col) order
            rowptr = torch.zeros(self.total_nodes + 1, dtype=torch.int64, device=dev)
            rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=self.total_nodes), 0)
            if int(rowptr[-1]) < 2 ** 31:
                rowptr = rowptr.to(torch.int32)
            self._csr = HopGraph.from_csr(rowptr, cols.to(torch.int32), codes, n_cols=self.total_nodes, n_codes=self.n_codes)
        return self._csr


def collate(batch):
    """Device-side counterpart of ``distance_collate_fn`` (batched_pyg_main.py:54-91): ``batch`` is a list of
    ``(x [n_g, F], dist [n_g, n_g], y)`` device tensors; returns ``(x_batch, blocks, y_batch, batch_vector)`` where ``blocks``
    (:class:`HopBlocks`) stands where the reference's ``dist_batch`` does — ``model(x_batch, blocks, batch_vector)``."""
    xs, dists, ys = zip(*batch)
    blocks = HopBlocks.from_blocks(dists)
    return torch.cat(xs, dim=0), blocks, torch.cat([y.reshape(-1) for y in ys], dim=0), blocks.batch_vector()


def hop_graph_from_counts(dist: torch.Tensor) -> HopGraph:
    """Hop-coded CSR of the listed pairs (``dist >= 0``) of ANY raw hop-count matrix (not necessarily block-diagonal);
    ``-1`` pairs are simply absent.  Integer work on the device (torch index ops over the dense matrix); bit-exact."""
    _lib.require_device(dist)
    if dist.dim() != 2 or dist.shape[0] != dist.shape[1]:
        raise ValueError(f"dist_batch must be square, got {tuple(dist.shape)}")
    n = dist.shape[0]
    listed = dist >= 0
    rows, cols = torch.nonzero(listed, as_tuple=True)            # row-major order == CSR order
    hops = dist[rows, cols]
    codes = hops.round()
    max_hop = int(codes.max()) if codes.numel() else 0
    if codes.numel() and (not bool((codes == hops).all()) or max_hop > 254):
        raise _lib.GnanHipError("dist_batch must hold integer hop counts in [0, 254] or -1")
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dist.device)
    rowptr[1:] = torch.cumsum(listed.sum(dim=1), 0)
    if int(rowptr[-1]) < 2 ** 31:
        rowptr = rowptr.to(torch.int32)
    return HopGraph.from_csr(rowptr, cols.to(torch.int32), codes.to(torch.uint8), n_cols=n, n_codes=max_hop + 2)


BATCH_BACKWARD_KERNEL = True             # the backward of a small batch in two launches (gnan_small_batch_bwd)
BATCH_BACKWARD_MAX_WORKSPACE = 256 << 20   # ... while the graphs' gradient slabs stay below 256 MiB


class _BatchedGraphs(torch.autograd.Function):
    """``[G, C]`` per-graph read-outs (or ``[N, C]`` node outputs) of a batch by ONE launch; backward: two launches
    (``gnan_small_batch_bwd``: every graph's share of the gradients, then their sum in graph order) for at most 64 hop codes,
    else the general kernels on the blocks' CSR from the saved node sums and rho table."""

    @staticmethod
    def forward(ctx, x, blocks: HopBlocks, graph_sum, fm, rm, cnt, raw_hops, *params):
        """``cnt`` (``[total_nodes, >= D]`` int32 shell sizes) and ``raw_hops=False``: the semantics of models.py:358-384 instead
        of the batched script's — rho on ``1 / (1 + hop)``, rho(0) on the unlisted pairs, weights divided by the shell size
        (``small_graph.SlotGraph``: a batch-size-1 loop through ONE captured step whatever the graphs' sizes)."""
        from . import functional as Fn
        Lf, Hf, Cf, F = fm
        Lr, Hr, Cr = rm
        xk = Fn._rows(x.detach().float())
        dev = xk.device
        G, N, D = blocks.n_graphs, blocks.total_nodes, blocks.n_codes
        keep_f = [None if t is None else Fn._c(t.detach()) for t in params[:6]]
        keep_r = [None if t is None else Fn._c(t.detach()) for t in params[6:]]
        S = torch.empty((N, Cf), dtype=torch.float32, device=dev)
        lut = torch.empty((G, D, Cr), dtype=torch.float32, device=dev)
        Y = None if graph_sum else torch.empty((N, Cf), dtype=torch.float32, device=dev)
        Ysum = torch.empty((G, Cf), dtype=torch.float32, device=dev) if graph_sum else None
        need = _lib.lib().gnan_small_batch_workspace_bytes(G, N, F, Cf)
        ws = torch.empty(need // 4 + 1, dtype=torch.int32, device=dev)
        ws[: 32 * G].zero_()                                                 # the graphs' arrival counters (a 128-byte line each)
        a = _lib.SmallBatchArgs(x=_lib.ptr(xk), x_stride=xk.stride(0), total_nodes=N, F=F, n_graphs=G,
                                max_nodes=blocks.max_nodes, f=_small_mlp(keep_f, Lf, Hf, Cf),
                                rho=_small_mlp(keep_r, Lr, Hr, Cr), code=_lib.ptr(blocks.code),
                                node_off=_lib.ptr(blocks.node_off), code_off=_lib.ptr(blocks.code_off), D=D,
                                rho_raw_hops=int(raw_hops), rest_zero=int(raw_hops), S=_lib.ptr(S), lut=_lib.ptr(lut),
                                Y=_lib.ptr(Y), Ysum=_lib.ptr(Ysum), workspace=_lib.ptr(ws), workspace_bytes=ws.numel() * 4,
                                cnt=_lib.ptr(cnt), cnt_stride=0 if cnt is None else cnt.stride(0))
        _lib.check(_lib.lib().gnan_small_batch_fwd(a, _lib.stream_of(xk)), "gnan_small_batch_fwd")
        ctx.blocks, ctx.graph_sum, ctx.fm, ctx.rm = blocks, graph_sum, fm, rm
        ctx.cnt, ctx.raw_hops = cnt, bool(raw_hops)
        ctx.present = [t is not None for t in params]
        ctx.dests = Fn._grad_dests_of(params)
        ctx.save_for_backward(xk, S, lut, *[t for t in params if t is not None])
        return Ysum if graph_sum else Y

    @staticmethod
    def backward(ctx, d_out):
        from . import functional as Fn
        saved = list(ctx.saved_tensors)
        x, S, lut = saved[:3]
        rest = saved[3:]
        params = [rest.pop(0) if pr else None for pr in ctx.present]
        Lf, Hf, Cf, F = ctx.fm
        Lr, Hr, Cr = ctx.rm
        need_f, need_r = any(ctx.needs_input_grad[7:13]), any(ctx.needs_input_grad[13:])
        blocks = ctx.blocks
        if (BATCH_BACKWARD_KERNEL and Cr in (1, Cf) and blocks.n_codes <= 64 and d_out.dtype == torch.float32
                and all(t is None or t.dtype == torch.float32 for t in params)):
            # two launches: every graph's workgroups leave its share of the gradients in a slab, the slabs are added in order
            keep = [None if t is None else Fn._c(t.detach()) for t in params]
            outs_f, outs_r = Fn._grad_outputs(keep[:6], ctx.dests[:6]), Fn._grad_outputs(keep[6:], ctx.dests[6:])

            def grads(o):
                return _lib.SmallMlpGrads(w_first=_lib.ptr(o[0]), b_first=_lib.ptr(o[1]),
                                          w_mid=None if o[2] is None else _lib.ptr(o[2][0]),
                                          b_mid=None if o[3] is None else _lib.ptr(o[3][0]), w_last=_lib.ptr(o[4]), b_last=_lib.ptr(o[5]))
            g_out = d_out.detach().contiguous()
            a = _lib.SmallBatchBwdArgs(x=_lib.ptr(x), x_stride=x.stride(0), total_nodes=blocks.total_nodes, F=F,
                                       n_graphs=blocks.n_graphs, max_nodes=blocks.max_nodes, f=_small_mlp(keep[:6], Lf, Hf, Cf),
                                       rho=_small_mlp(keep[6:], Lr, Hr, Cr), code=_lib.ptr(blocks.code),
                                       node_off=_lib.ptr(blocks.node_off), code_off=_lib.ptr(blocks.code_off), D=blocks.n_codes,
                                       rho_raw_hops=int(ctx.raw_hops), rest_zero=int(ctx.raw_hops), S=_lib.ptr(S), lut=_lib.ptr(lut),
                                       cnt=_lib.ptr(ctx.cnt), cnt_stride=0 if ctx.cnt is None else ctx.cnt.stride(0),
                                       dY=None if ctx.graph_sum else _lib.ptr(g_out),
                                       dYsum=_lib.ptr(g_out) if ctx.graph_sum else None, df=grads(outs_f), drho=grads(outs_r),
                                       workspace=None, workspace_bytes=0)
            need = _lib.lib().gnan_small_batch_bwd_workspace_bytes(a)
            if need <= BATCH_BACKWARD_MAX_WORKSPACE:
                ws = torch.empty(need // 4 + 1, dtype=torch.float32, device=x.device)
                a.workspace, a.workspace_bytes = _lib.ptr(ws), ws.numel() * 4
                _lib.check(_lib.lib().gnan_small_batch_bwd(a, _lib.stream_of(x)), "gnan_small_batch_bwd")
                return (None, None, None, None, None, None, None, *[o if need_f else None for o in outs_f],
                        *[o if need_r else None for o in outs_r])
        if ctx.cnt is not None or not ctx.raw_hops:
            raise _lib.GnanHipError("gnan_small_batch_bwd does not cover this slot step (more than 64 hop codes, a rho of several "
                                    "channels or too large a workspace): no other route reads the slots' sizes from the device")
        dY = d_out.index_select(0, blocks.batch_vector()) if ctx.graph_sum else d_out      # every node gets its graph's gradient
        from . import aggregate
        bag = aggregate._Bag()
        bag.g, bag.use_cnt, bag.with_rest, bag.row_ids, bag.reduce_cr = blocks.csr(), False, False, None, 0
        bag.s_total, bag.total_rows, bag.total_group = None, None, aggregate.NOT_SHARED
        dS, dlut = aggregate._aggregate_backward(bag, S, lut[0].contiguous(), dY.contiguous(), need_f, need_r)
        pg_f = pg_r = [None] * 6
        if need_f:
            _, pg_f = Fn._shape_function_grads(x, params[:6], ctx.present[:6], None, dS, True, Lf, Hf, Cf, F, dests=ctx.dests[:6])
        if need_r:
            D = blocks.n_codes
            hops = torch.arange(D - 1, dtype=torch.float32, device=x.device).view(-1, 1)     # rho's inputs: the raw hop counts
            _, pg_r = Fn._shape_function_grads(hops, params[6:], ctx.present[6:], None, dlut[: D - 1].reshape(-1, Cr), False,
                                               Lr, Hr, Cr, 1, dests=ctx.dests[6:])
        return (None, None, None, None, None, None, None, *pg_f, *pg_r)


class TensorGNAN(_PathBase):
    """``TensorGNAN`` of the batched script — constructor batched_pyg_main.py:99-131, forward :133-184.
    The shape functions and rho are always two layers deep there (``n_layers`` is accepted and unused).
    ``dist_batch`` may be the reference's dense matrix or the :class:`HopBlocks` of :func:`collate`."""

    def __init__(self, in_channels, out_channels, n_layers, hidden_channels=16, device='cpu',
                 bias=True, dropout=0.0, is_graph_task=True):
        super().__init__()
        self.device = device
        self.out_channels = out_channels
        self.is_graph_task = is_graph_task
        self.dropout = dropout
        self.fs = nn.ModuleList(_two_layer(hidden_channels, out_channels, bias, dropout) for _ in range(in_channels))
        self.rho = _two_layer(hidden_channels, out_channels, bias, dropout)
        self._init_caches()

    def _blocks(self, dist_batch, batch_vector) -> Optional[HopBlocks]:
        if isinstance(dist_batch, HopBlocks):
            return dist_batch
        hit = self._graph_cache.get((dist_batch, batch_vector), "blocks")
        if hit is None:
            try:
                hit = HopBlocks.from_dense(dist_batch, batch_vector)
            except _lib.GnanHipError:
                hit = False                              # not block-diagonal / unsorted: the general CSR of the whole matrix
            self._graph_cache.put((dist_batch, batch_vector), "blocks", hit)
        return hit or None

    def forward(self, x_batch, dist_batch, batch_vector):
        self._check_dropout()
        _lib.require_device(x_batch)
        blocks = self._blocks(dist_batch, batch_vector)
        f, rho = self._stacked("fs", self.fs), self._stacked("rho", [self.rho])
        # (a graph without nodes — a gap in batch_vector's ids — has no workgroup to write its row: the CSR route gives it zeros)
        if (BATCH_KERNEL and blocks is not None and not self._dropout_active() and 1 <= blocks.min_nodes
                and blocks.max_nodes <= BATCH_KERNEL_MAX_NODES
                and blocks.total_nodes <= BATCH_KERNEL_MAX_TOTAL_NODES and blocks.n_codes <= 256 and f.H <= 64 and f.C <= 8 and x_batch.dtype == torch.float32
                and not x_batch.requires_grad and blocks.n_graphs <= 65535):
            fm, rm = (f.L, f.H, f.C, f.F), (rho.L, rho.H, rho.C)
            return _BatchedGraphs.apply(x_batch, blocks, bool(self.is_graph_task), fm, rm, None, True, *f[:6], *rho[:6])
        if blocks is not None:
            g = blocks.csr()
        else:
            _lib.require_device(dist_batch)
            g = self._graph_cache.get((dist_batch,), "counts")
            if g is None:
                g = self._graph_cache.put((dist_batch,), "counts", hop_graph_from_counts(dist_batch))
        S = self._features(x_batch, "fs", self.fs, True)                                    # [N, C]
        hops = torch.arange(g.n_codes - 1, dtype=torch.float32, device=x_batch.device).view(-1, 1)
        if self._dropout_active():
            # rho carries a Dropout here (batched_pyg_main.py:126-131) and the reference draws one mask per PAIR
            # (rho runs on all N*N distances, :151).  A table of rho at the distinct hop counts would share one mask
            # among all pairs of a hop count, so while Dropout is active rho is evaluated per listed pair and the
            # weighted sum runs in torch on the device — the cold path, like the shape functions' (_check_dropout).
            deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
            row_of_pair = torch.repeat_interleave(torch.arange(g.n_rows, device=deg.device), deg)
            self._stores["rho"].direct_use()                                                # (the module itself joins the autograd graph)
            w_pair = self.rho(hops[g.code.long()])                                          # [nnz, C], one mask per pair
            Y = torch.zeros_like(S).index_add(0, row_of_pair, w_pair * S[g.col.long()])
        else:
            # rho on the handful of distinct hop counts: one launch of the shape-function kernel (one of gnan_fmlp_bwd in the
            # backward pass) instead of the six framework launches of self.rho(hops), as modules._lut_global
            from .functional import feature_mlps
            listed = feature_mlps(hops, rho, sum_features=False)
            lut = torch.cat([listed, torch.zeros(1, self.out_channels, device=x_batch.device)], dim=0)
            Y = rho_aggregate(g, S, lut, use_cnt=False, with_rest=False)                    # [N, C]
        if not self.is_graph_task:
            return Y
        if blocks is not None and Y.shape[1] <= 64:
            return _SegmentSum.apply(Y, blocks)                                             # batched_pyg_main.py:176-181
        bv = batch_vector.to(Y.device).long()               # a batch that is not block-diagonal: the reference's own scatter
        out = torch.zeros(int(bv.max()) + 1, Y.shape[1], device=Y.device, dtype=Y.dtype)
        return out.index_add(0, bv, Y)


class GraphedBatchStep:
    """A captured step for batches of graphs (batched_pyg_main.py:205-226: forward, loss, backward, optimizer step), replayed
    from ONE hipGraph launch per batch.

    Every batch has another shape, so the step is captured over SLOTS: feature rows ``[node_capacity, F]``, packed hop codes,
    the two offset arrays and the labels.  The kernels of a small batch read every graph's size from the offset arrays on the
    device (``gnan_small_batch_fwd`` / ``_bwd``: blockIdx.y = graph), so one capture serves any batch of ``n_graphs`` graphs
    that fits the slots; ``n_codes`` is captured at a capacity too (hop codes beyond a batch's own largest hop are the rest
    code's business: weight 0, no gradient).  :meth:`run` copies a batch into the slots (one launch) and replays; it returns
    None for a batch that does not fit (another number of graphs, more nodes, a longer hop): the caller steps eagerly.

    ``loss_of(outputs, labels) -> loss`` — or a torch loss module: ``BCEWithLogitsLoss`` / ``CrossEntropyLoss`` with mean
    reduction then run as ONE launch (``losses.loss_step``: loss and its gradient together) instead of torch's five;
    ``optimizer=None`` captures the evaluation pass.  Construction runs the FIRST batch's step eagerly (a real step: the
    parameters are updated) and captures after it.  ``prepared``: the optimizer's capturable mode if another captured step of
    the same optimizer switched it on already (``other.step.prepared``) — two steps that each prepare the optimizer
    invalidate one another.
    """

    def __init__(self, model: "TensorGNAN", optimizer, loss_of, x, blocks: HopBlocks, labels, node_capacity: Optional[int] = None,
                 n_codes: Optional[int] = None, prepared=None, label_flag: Optional[torch.Tensor] = None):
        from .graphed import GraphedStep
        _lib.require_device(x, labels)
        # the labels change with every replayed batch and nobody looks at them on the host: the fused cross entropy raises this
        # flag on the device for a class label outside [0, C) (torch would have left such a row out of the mean); whoever reads
        # the epoch's totals reads it too (train_epoch below)
        self.label_flag = label_flag if label_flag is not None else torch.zeros(1, dtype=torch.float32, device=x.device)
        if isinstance(loss_of, nn.Module):
            from .losses import loss_kind, loss_step
            loss_module = loss_of

            def loss_of(out, lab):                   # noqa: F811  (the module's loss, fused where the kernel covers it)
                kind = loss_kind(loss_module, out)
                if kind is not None and out.shape[0] == lab.numel() and lab.dim() == 1:
                    return loss_step(out, lab, kind, want_hits=False, unit_upstream=optimizer is not None,
                                     label_flag=self.label_flag)[0]
                return loss_module(out, lab)
        dev = x.device
        self.n_graphs, self.F = blocks.n_graphs, int(x.shape[1])
        self.node_capacity = int(node_capacity or BATCH_KERNEL_MAX_TOTAL_NODES)
        self.n_codes = int(n_codes or min(64, max(16, blocks.n_codes + 8)))
        self.code_capacity = self.n_graphs * BATCH_KERNEL_MAX_NODES * BATCH_KERNEL_MAX_NODES
        if not self.fits(x, blocks, labels):
            raise ValueError("the first batch does not fit the slots it defines")
        self.x = torch.zeros((self.node_capacity, self.F), dtype=torch.float32, device=dev)
        self.code = torch.full((self.code_capacity,), 255, dtype=torch.uint8, device=dev)
        self.node_off = torch.zeros(self.n_graphs + 1, dtype=torch.int32, device=dev)
        self.code_off = torch.zeros(self.n_graphs + 1, dtype=torch.int64, device=dev)
        self.labels = torch.empty_like(labels)
        self.blocks = HopBlocks(self.code, self.node_off, self.code_off, [0] * self.n_graphs, self.n_codes - 2)
        self.blocks.total_nodes, self.blocks.max_nodes, self.blocks.slots = self.node_capacity, BATCH_KERNEL_MAX_NODES, True
        self.blocks.min_nodes = 1                            # (fits() admits batches of non-empty graphs only)
        self._labels_shape = tuple(labels.shape)
        self.load(x, blocks, labels)
        self.step = GraphedStep(model, None, lambda out: (loss_of(out, self.labels), None), optimizer,
                                forward=lambda: model(self.x, self.blocks, None), warmup=1, prepared=prepared)

    def fits(self, x, blocks: HopBlocks, labels) -> bool:
        return bool(blocks.n_graphs == self.n_graphs and x.shape[1] == self.F and blocks.total_nodes <= self.node_capacity
                    and blocks.min_nodes >= 1
                    and blocks.n_codes <= self.n_codes and blocks.max_nodes <= BATCH_KERNEL_MAX_NODES
                    and int(blocks.code.numel()) <= self.code_capacity and x.dtype == torch.float32 and x.is_cuda
                    and (not hasattr(self, "_labels_shape") or tuple(labels.shape) == self._labels_shape))

    def load(self, x, blocks: HopBlocks, labels) -> None:
        n, m = int(x.shape[0]), int(blocks.code.numel())
        pairs = [(self.x[:n], x), (self.code[:m], blocks.code), (self.node_off, blocks.node_off), (self.code_off, blocks.code_off),
                 (self.labels, labels)]
        if all(s.is_contiguous() and d.is_contiguous() and s.dtype == d.dtype and s.shape == d.shape and s.is_cuda for d, s in pairs):
            _lib.multi_copy(pairs)                    # one launch
            return
        for d, s in pairs:
            d.copy_(s)

    def run(self, x, blocks: HopBlocks, labels):
        """``(outputs [G, C], loss, None)`` of the replayed step on this batch (static tensors, overwritten by the next replay),
        or None if the batch does not fit the slots or the capture has gone stale."""
        if not self.fits(x, blocks, labels) or self.step.graph is None or self.step.stale():
            return None
        self.load(x, blocks, labels)
        return self.step.replay()

    @property
    def kernel_nodes(self) -> int:
        return int(self.step.graph.kernel_nodes)


def train_epoch(model: "TensorGNAN", loader, loss_fn, optimizer, steps: Optional[dict] = None):
    """One pass of the script's training loop (batched_pyg_main.py:205-226): per batch ``zero_grad``, forward, loss, accuracy
    (``argmax == label``), ``backward``, ``optimizer.step()``; returns ``(mean loss, mean accuracy, steps)`` — the means over
    the batches, as the script prints them.  ``loader`` yields ``(x_batch, dist_batch or HopBlocks, y_batch, batch_vector)``
    (:func:`collate`'s tuples).  Batches that fit are replayed from a captured step (:class:`GraphedBatchStep`, one per number
    of graphs, kept in ``steps`` — hand the returned dict to the next epoch); the others (a last, smaller batch; a batch with
    more nodes than the slots hold) are stepped eagerly.  Loss and hit counts are added up on the device and read once."""
    steps = {} if steps is None else steps
    dev = next(model.parameters()).device
    totals = torch.zeros(2, dtype=torch.float64, device=dev)            # sum of batch losses, sum of batch accuracies
    n_batches = 0
    for x, dist, y, bv in loader:
        blocks = dist if isinstance(dist, HopBlocks) else model._blocks(dist, bv)
        out = None
        if blocks is not None and GRAPHED_BATCH_STEPS and not model._dropout_active():
            # a captured step belongs to the (model, loss, optimizer) it was captured with: another loss callable or optimizer
            # handed in with the same dict starts over
            owner = (id(model), id(loss_fn), id(optimizer))
            if steps.get("owner") != owner:
                for k in [k for k in steps if k != "owner"]:
                    old = steps.pop(k)
                    if isinstance(old, GraphedBatchStep):
                        old.step.release(restore_optimizer=True)
                steps["owner"] = owner
            gs = steps.get(blocks.n_graphs)
            if (gs is None and blocks.total_nodes <= BATCH_KERNEL_MAX_TOTAL_NODES and blocks.max_nodes <= BATCH_KERNEL_MAX_NODES
                    and blocks.min_nodes >= 1):
                from .graphed import CaptureFailed
                try:
                    if "label_flag" not in steps:
                        steps["label_flag"] = torch.zeros(1, dtype=torch.float32, device=dev)
                    gs = steps[blocks.n_graphs] = GraphedBatchStep(model, optimizer, loss_fn, x, blocks, y,
                                                                   prepared=steps.get("prepared"), label_flag=steps["label_flag"])
                    steps["prepared"] = gs.step.prepared
                    out, loss = gs.step.warmup_result[0], gs.step.warmup_result[1]     # (construction ran this batch's step)
                except (CaptureFailed, ValueError) as e:
                    # not capturable (a loss callable that synchronises, an optimizer without a capturable mode, a first batch
                    # that does not fit its own slots): this number of graphs stays on the eager loop.  Anything else — out of
                    # memory, an ABI mismatch — is the caller's to see.  If the construction already STEPPED this batch (its
                    # warm-up step is a real one) that step is this batch's: stepping it again would update twice.
                    import warnings
                    warnings.warn(f"gnan_amd.batched: batches of {blocks.n_graphs} graphs stay on the eager loop ({e})")
                    steps[blocks.n_graphs] = gs = False
                    done = getattr(e, "warmup_result", None)
                    if done is not None:
                        out, loss = done[0], done[1]
            elif gs:
                got = gs.run(x, blocks, y)
                if got is not None:
                    out, loss = got[0], got[1]
        if out is None:
            optimizer.zero_grad(set_to_none=True)
            out = model(x, dist, bv)
            loss = loss_fn(out, y)
            loss.backward()
            optimizer.step()
        with torch.no_grad():
            totals[0] += loss.detach().double()
            totals[1] += (out.detach().argmax(dim=-1) == y).double().mean()
        n_batches += 1
    flag = steps.get("label_flag")
    mean = (totals / max(n_batches, 1)).tolist() if flag is None else torch.cat([totals / max(n_batches, 1), flag.double()]).tolist()
    if flag is not None and mean[2] != 0.0:
        flag.zero_()
        raise _lib.GnanHipError("a replayed batch held a class label outside [0, C): the fused cross entropy averages over such rows, "
                                "torch's leaves them out (ignore_index) — this epoch's steps are not the script's; set "
                                "gnan_amd.batched.GRAPHED_BATCH_STEPS = False for data with ignored labels")
    return mean[0], mean[1], steps


GRAPHED_BATCH_STEPS = True       # batched.train_epoch replays captured steps where a batch fits (off: the eager loop)
