"""One-launch forward and backward of SMALL dense-coded graphs (``csrc/small_graph.hip``, ``csrc/small_graph_nam.hip``) — what a
graph-level task feeds per step (trainer.py:23-86 with ``batch_size = 1``; SURVEY.md section 8d C2: 30 nodes on average).

* :func:`small_graph_forward` — features summed per node, rho on the distinct distances (or, ``use_cnt="pre"``, on the n x D
  normalised distances of GNAN.py:65-67), the normalised aggregation and the graph read-out: GNAN.py:55-79 / 146-172,
  models.py:358-384 with ``readout_n_layers == 0``;
* :func:`small_graph_nam_forward` — the NAM read-out over the per-feature aggregates (models.py:379-381).

Both are autograd nodes whose backward pass is one launch as well.  The general kernels (``functional``) take everything these
do not cover; the ``*_applies`` predicates say which is which.
"""
from __future__ import annotations

import torch

from . import _lib
from . import aggregate
from . import functional as Fn
from .functional import StackedMLP
from .graph import HopGraph, hop_inputs

SMALL_GRAPH_FORWARD = True
SMALL_GRAPH_MAX_NODES = 128   # gnan_small_graph_fwd / _bwd: one block of 64 nodes (static LDS) or two (64 KB of tables)
SMALL_GRAPH_BACKWARD = True   # ... and its backward pass (gnan_small_graph_bwd)
_SMALL_WS = {}               # (device index, stream) -> workspace whose counter word the kernel leaves zero


def small_graph_applies(x: torch.Tensor, g: HopGraph, f: StackedMLP, rho: StackedMLP, pre_rho: bool = False) -> bool:
    """Can ``gnan_small_graph_fwd`` take this forward (features summed per node; post-rho normalisation, or — ``pre_rho`` — the
    stand-alone file's rho(distance / shell size), GNAN.py:65-67, with a one-channel rho and at most 64 shells)?"""
    if pre_rho and not (g.cnt is not None and rho.C == 1 and g.n_codes <= 64 and SMALL_GRAPH_BACKWARD):
        return False
    return bool(SMALL_GRAPH_FORWARD and g.is_dense and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad
                and 1 <= x.shape[0] <= SMALL_GRAPH_MAX_NODES and g.n_rows == g.n_cols == x.shape[0] and g.n_codes <= 256
                and x.shape[1] == f.F and rho.F == 1 and f.L in (2, 3) and rho.L in (2, 3) and 1 <= f.H <= 64
                and 1 <= rho.H <= 64 and f.C <= 8 and rho.C in (1, f.C)
                and all(t is None or t.dtype == torch.float32 for t in tuple(f[:6]) + tuple(rho[:6])))


def _small_mlp(keep, L, H, C) -> "_lib.SmallMlp":
    w_mid = None if keep[2] is None else keep[2][0]
    b_mid = None if keep[3] is None else keep[3][0]
    return _lib.SmallMlp(L=L, H=H, C=C, w_first=_lib.ptr(keep[0]), b_first=_lib.ptr(keep[1]), w_mid=_lib.ptr(w_mid),
                         b_mid=_lib.ptr(b_mid), w_last=_lib.ptr(keep[4]), b_last=_lib.ptr(keep[5]))


class _SmallGraph(torch.autograd.Function):
    """``Y`` (``[n, C]``) or its sum over the nodes (``[C, 1]``, ``graph_sum``) by ONE launch; backward: one launch too
    (``gnan_small_graph_bwd``: one rho channel, <= 64 shells), else the general path's kernels on the saved node sums and rho
    table — transposed aggregation, table gradient, the two small-batch MLP backward launches (their gradients land in the flat
    gradient buffers directly).  ``use_cnt``: False / True (post-rho shell normalisation) / ``"pre"`` (GNAN.py:65-67)."""

    @staticmethod
    def forward(ctx, x, g, use_cnt, graph_sum, fm, rm, *params):
        Lf, Hf, Cf, F = fm
        Lr, Hr, Cr = rm
        fp, rp = params[:6], params[6:]
        xk = Fn._rows(x.detach())
        n, D = xk.shape[0], g.n_codes
        dev = xk.device
        keep_f = [None if t is None else Fn._c(t.detach()) for t in fp]
        keep_r = [None if t is None else Fn._c(t.detach()) for t in rp]
        pre_rho = use_cnt == "pre"
        S = torch.empty((n, Cf), dtype=torch.float32, device=dev)
        lut = torch.empty((n * D if pre_rho else D, Cr), dtype=torch.float32, device=dev)
        Y = None if graph_sum else torch.empty((n, Cf), dtype=torch.float32, device=dev)
        Ysum = torch.empty(Cf, dtype=torch.float32, device=dev) if graph_sum else None
        need = _lib.lib().gnan_small_graph_workspace_bytes(n, F, Cf)
        if torch.cuda.is_current_stream_capturing():
            # a captured step owns its workspace: zeroed eagerly BEFORE the capture (graphed.GraphedCallable) — every capture
            # runs on the framework's one capture stream, so a buffer keyed by stream would be shared by all captured steps,
            # and one allocated inside a capture would have its zero fill recorded, never run
            ws = Fn.CAPTURE_SCRATCH
            if ws is None or (ws.numel() - Fn.ARRIVE_SLOTS) * 4 < need:      # (its last words are arrival counters)
                ws = torch.zeros(need // 4 + 1, dtype=torch.int32, device=dev)     # (recorded fill: runs at every replay)
        else:
            key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
            ws = _SMALL_WS.get(key)
            if ws is None or ws.numel() * 4 < need:
                ws = _SMALL_WS[key] = torch.zeros(max(need // 4 + 1, 1 << 16), dtype=torch.int32, device=dev)
        cnt = g.cnt if use_cnt else None
        a = _lib.SmallGraphArgs(x=_lib.ptr(xk), x_stride=xk.stride(0), n=n, F=F, f=_small_mlp(keep_f, Lf, Hf, Cf),
                                rho=_small_mlp(keep_r, Lr, Hr, Cr), code=_lib.ptr(g.code), D=D, pre_rho=int(pre_rho),
                                cnt=_lib.ptr(cnt), cnt_stride=0 if cnt is None else cnt.stride(0), S=_lib.ptr(S),
                                lut=_lib.ptr(lut), Y=_lib.ptr(Y),
                                Ysum=_lib.ptr(Ysum), workspace=_lib.ptr(ws), workspace_bytes=ws.numel() * 4)
        _lib.check(_lib.lib().gnan_small_graph_fwd(a, _lib.stream_of(xk)), "gnan_small_graph_fwd")
        ctx.g, ctx.use_cnt, ctx.graph_sum, ctx.fm, ctx.rm = g, use_cnt, graph_sum, fm, rm
        ctx.present = [t is not None for t in params]
        ctx.dests = Fn._grad_dests_of(params)
        ctx.save_for_backward(xk, S, lut, *[t for t in params if t is not None])
        return Ysum.view(-1, 1) if graph_sum else Y

    @staticmethod
    def backward(ctx, d_out):
        saved = list(ctx.saved_tensors)
        x, S, lut = saved[:3]
        rest = saved[3:]
        params = [rest.pop(0) if pr else None for pr in ctx.present]
        fp, rp = params[:6], params[6:]
        Lf, Hf, Cf, F = ctx.fm
        Lr, Hr, Cr = ctx.rm
        n = x.shape[0]
        need_f, need_r = any(ctx.needs_input_grad[6:12]), any(ctx.needs_input_grad[12:])
        pre_rho = ctx.use_cnt == "pre"           # (small_graph_applies admitted it only where the one launch below covers it)
        if ((SMALL_GRAPH_BACKWARD and Cr == 1 and ctx.g.n_codes <= 64 and d_out.dtype == torch.float32
                and all(t is None or t.dtype == torch.float32 for t in params)) and ((need_f and need_r) or pre_rho)):
            # one launch: every workgroup forms the operand (or table) gradient it needs itself, then the small-batch MLP backward
            keep = [None if t is None else Fn._c(t.detach()) for t in params]
            outs_f, outs_r = Fn._grad_outputs(keep[:6], ctx.dests[:6]), Fn._grad_outputs(keep[6:], ctx.dests[6:])

            def grads(o):
                return _lib.SmallMlpGrads(w_first=_lib.ptr(o[0]), b_first=_lib.ptr(o[1]),
                                          w_mid=None if o[2] is None else _lib.ptr(o[2][0]),
                                          b_mid=None if o[3] is None else _lib.ptr(o[3][0]), w_last=_lib.ptr(o[4]), b_last=_lib.ptr(o[5]))
            g_out = d_out.detach().contiguous()
            cnt = ctx.g.cnt if ctx.use_cnt else None
            ws = _workspace(x.device, _lib.lib().gnan_small_graph_bwd_workspace_bytes(n, ctx.g.n_codes)) if pre_rho else None
            a = _lib.SmallGraphBwdArgs(x=_lib.ptr(x), x_stride=x.stride(0), n=n, F=F, f=_small_mlp(keep[:6], Lf, Hf, Cf),
                                       rho=_small_mlp(keep[6:], Lr, Hr, Cr), code=_lib.ptr(ctx.g.code), D=ctx.g.n_codes,
                                       pre_rho=int(pre_rho), cnt=_lib.ptr(cnt), cnt_stride=0 if cnt is None else cnt.stride(0),
                                       S=_lib.ptr(S),
                                       lut=_lib.ptr(lut), dY=None if ctx.graph_sum else _lib.ptr(g_out),
                                       dYsum=_lib.ptr(g_out) if ctx.graph_sum else None, df=grads(outs_f), drho=grads(outs_r),
                                       workspace=_lib.ptr(ws), workspace_bytes=0 if ws is None else ws.numel() * 4)
            _lib.check(_lib.lib().gnan_small_graph_bwd(a, _lib.stream_of(x)), "gnan_small_graph_bwd")
            if not need_f:
                outs_f = [None] * 6
            if not need_r:
                outs_r = [None] * 6
            return (None, None, None, None, None, None, *outs_f, *outs_r)
        if pre_rho:
            raise RuntimeError("the one-launch small-graph forward with pre-rho normalisation was taken where its backward "
                               "does not apply (float32 parameters and output gradient expected)")
        dY = d_out.reshape(1, Cf).expand(n, Cf).contiguous() if ctx.graph_sum else d_out.contiguous()
        bag = aggregate._Bag()
        bag.g, bag.use_cnt, bag.with_rest, bag.row_ids, bag.reduce_cr = ctx.g, ctx.use_cnt, False, None, 0
        bag.s_total, bag.total_rows, bag.total_group = None, None, aggregate.NOT_SHARED
        dS, dlut = aggregate._aggregate_backward(bag, S, lut, dY, need_f, need_r)
        pg_f = pg_r = [None] * 6
        if need_f:
            _, pg_f = Fn._shape_function_grads(x, fp, ctx.present[:6], None, dS, True, Lf, Hf, Cf, F, dests=ctx.dests[:6])
        if need_r:
            u = hop_inputs(ctx.g.n_codes, x.device).view(-1, 1)
            _, pg_r = Fn._shape_function_grads(u, rp, ctx.present[6:], None, dlut.reshape(-1, Cr), False, Lr, Hr, Cr, 1,
                                            dests=ctx.dests[6:])
        return (None, None, None, None, None, None, *pg_f, *pg_r)


def small_graph_forward(x: torch.Tensor, g: HopGraph, f: StackedMLP, rho: StackedMLP, use_cnt, graph_sum: bool):
    """The forward of GNAN.py:146-172 / models.py:358-384 (post-rho; ``use_cnt="pre"``: GNAN.py:55-79, pre-rho) on a small
    dense-coded graph by ``gnan_small_graph_fwd``: ``[n, C]`` node outputs, or ``[C, 1]`` with ``graph_sum`` (GNAN.py:75-79)."""
    _lib.require_device(x, f.w_last, rho.w_last, g.code)
    return _SmallGraph.apply(x, g, "pre" if use_cnt == "pre" else bool(use_cnt), bool(graph_sum), (f.L, f.H, f.C, f.F),
                             (rho.L, rho.H, rho.C),
                             *f[:6], *rho[:6])


# =============================================================================
# ... with a NAM read-out over the per-feature aggregates (csrc/small_graph_nam.hip)
# =============================================================================
SMALL_GRAPH_NAM = True


def _workspace(dev, need: int) -> torch.Tensor:
    """The kernels' counter + scratch: zero-filled, left zero by every launch, one per (device, stream) — or the captured
    step's own (see ``_SmallGraph.forward``)."""
    if torch.cuda.is_current_stream_capturing():
        ws = Fn.CAPTURE_SCRATCH
        if ws is None or (ws.numel() - Fn.ARRIVE_SLOTS) * 4 < need:      # (its last words are arrival counters)
            ws = torch.zeros(need // 4 + 1, dtype=torch.int32, device=dev)
        return ws
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _SMALL_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = _SMALL_WS[key] = torch.zeros(max(need // 4 + 1, 1 << 16), dtype=torch.int32, device=dev)
    return ws


def _nam_mlp(keep, L, H, C) -> "_lib.SmallMlp":
    if L == 1:                                       # Linear(1, C) per feature: w_last [F, C]
        return _lib.SmallMlp(L=1, H=0, C=C, w_first=None, b_first=None, w_mid=None, b_mid=None, w_last=_lib.ptr(keep[4]),
                             b_last=_lib.ptr(keep[5]))
    return _small_mlp(keep, L, H, C)


SMALL_GRAPH_NAM_MAX_FEATURES = 127   # the backward's workgroups (one per feature + rho's) must be resident together (csrc/small_graph_nam.hip)


def small_graph_nam_applies(x: torch.Tensor, g: HopGraph, f: StackedMLP, rho: StackedMLP, nam: StackedMLP) -> bool:
    """Can ``gnan_small_graph_nam_fwd`` / ``_bwd`` take this graph-level forward with a NAM read-out?"""
    stacks = tuple(f[:6]) + tuple(rho[:6]) + tuple(nam[:6])
    return bool(SMALL_GRAPH_NAM and g.is_dense and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad
                and 1 <= x.shape[0] <= SMALL_GRAPH_MAX_NODES and g.n_rows == g.n_cols == x.shape[0] and g.n_codes <= 64
                and x.shape[1] == f.F == nam.F and f.F <= SMALL_GRAPH_NAM_MAX_FEATURES and rho.F == 1 and f.C == 1 and rho.C == 1 and f.L in (2, 3) and rho.L in (2, 3)
                and nam.L in (1, 2, 3) and 1 <= f.H <= 64 and 1 <= rho.H <= 64 and (nam.L == 1 or 1 <= nam.H <= 64)
                and 1 <= nam.C <= 8 and all(t is None or t.dtype == torch.float32 for t in stacks))


class _SmallGraphNam(torch.autograd.Function):
    """``out [C, 1] = sum_k nam_k(sum_ij w_ij f_k(x_jk))`` by ONE launch; backward: one launch as well."""

    @staticmethod
    def forward(ctx, x, g, use_cnt, fm, rm, nm, *params):
        Lf, Hf, F = fm
        Lr, Hr = rm
        Ln, Hn, Cn = nm
        xk = Fn._rows(x.detach())
        n, D, dev = xk.shape[0], g.n_codes, xk.device
        keep = [None if t is None else Fn._c(t.detach()) for t in params]
        fx = torch.empty((F, n), dtype=torch.float32, device=dev)
        lut = torch.empty(D, dtype=torch.float32, device=dev)
        hidden = torch.empty(F, dtype=torch.float32, device=dev)
        out = torch.empty(Cn, dtype=torch.float32, device=dev)
        ws = _workspace(dev, _lib.lib().gnan_small_graph_nam_workspace_bytes(F, Cn))
        cnt = g.cnt if use_cnt else None
        a = _lib.SmallGraphNamArgs(x=_lib.ptr(xk), x_stride=xk.stride(0), n=n, F=F, f=_small_mlp(keep[:6], Lf, Hf, 1),
                                   rho=_small_mlp(keep[6:12], Lr, Hr, 1), nam=_nam_mlp(keep[12:], Ln, Hn, Cn),
                                   code=_lib.ptr(g.code), D=D, reserved=0, cnt=_lib.ptr(cnt),
                                   cnt_stride=0 if cnt is None else cnt.stride(0), fx=_lib.ptr(fx), lut=_lib.ptr(lut),
                                   hidden=_lib.ptr(hidden), out=_lib.ptr(out), workspace=_lib.ptr(ws), workspace_bytes=ws.numel() * 4)
        _lib.check(_lib.lib().gnan_small_graph_nam_fwd(a, _lib.stream_of(xk)), "gnan_small_graph_nam_fwd")
        ctx.g, ctx.use_cnt, ctx.fm, ctx.rm, ctx.nm = g, use_cnt, fm, rm, nm
        ctx.present = [t is not None for t in params]
        ctx.dests = Fn._grad_dests_of(params)
        ctx.save_for_backward(xk, fx, lut, hidden, *[t for t in params if t is not None])
        return out.view(-1, 1)

    @staticmethod
    def backward(ctx, d_out):
        saved = list(ctx.saved_tensors)
        x, fx, lut, hidden = saved[:4]
        rest = saved[4:]
        params = [rest.pop(0) if pr else None for pr in ctx.present]
        Lf, Hf, F = ctx.fm
        Lr, Hr = ctx.rm
        Ln, Hn, Cn = ctx.nm
        n, dev = x.shape[0], x.device
        keep = [None if t is None else Fn._c(t.detach()) for t in params]
        outs = [Fn._grad_outputs(keep[i:i + 6], ctx.dests[i:i + 6]) for i in (0, 6, 12)]

        def grads(o, L):
            if L == 1:
                return _lib.SmallMlpGrads(w_first=None, b_first=None, w_mid=None, b_mid=None, w_last=_lib.ptr(o[4]), b_last=_lib.ptr(o[5]))
            return _lib.SmallMlpGrads(w_first=_lib.ptr(o[0]), b_first=_lib.ptr(o[1]), w_mid=None if o[2] is None else _lib.ptr(o[2][0]),
                                      b_mid=None if o[3] is None else _lib.ptr(o[3][0]), w_last=_lib.ptr(o[4]), b_last=_lib.ptr(o[5]))
        g_out = d_out.detach().to(torch.float32).contiguous().view(-1)
        ws = _workspace(dev, _lib.lib().gnan_small_graph_nam_workspace_bytes(F, Cn))
        cnt = ctx.g.cnt if ctx.use_cnt else None
        a = _lib.SmallGraphNamBwdArgs(x=_lib.ptr(x), x_stride=x.stride(0), n=n, F=F, f=_small_mlp(keep[:6], Lf, Hf, 1),
                                      rho=_small_mlp(keep[6:12], Lr, Hr, 1), nam=_nam_mlp(keep[12:], Ln, Hn, Cn),
                                      code=_lib.ptr(ctx.g.code), D=ctx.g.n_codes, reserved=0, cnt=_lib.ptr(cnt),
                                      cnt_stride=0 if cnt is None else cnt.stride(0), fx=_lib.ptr(fx), lut=_lib.ptr(lut),
                                      hidden=_lib.ptr(hidden), d_out=_lib.ptr(g_out), df=grads(outs[0], Lf), drho=grads(outs[1], Lr),
                                      dnam=grads(outs[2], Ln), workspace=_lib.ptr(ws), workspace_bytes=ws.numel() * 4)
        _lib.check(_lib.lib().gnan_small_graph_nam_bwd(a, _lib.stream_of(x)), "gnan_small_graph_nam_bwd")
        flat = []
        for i, o in zip((0, 6, 12), outs):
            need = ctx.needs_input_grad[6 + i:12 + i]
            flat += [v if nd else None for v, nd in zip(o, need)]
        return (None, None, None, None, None, None, *flat)


def small_graph_nam_forward(x: torch.Tensor, g: HopGraph, f: StackedMLP, rho: StackedMLP, nam: StackedMLP, use_cnt: bool):
    """models.py:358-384 with ``is_graph_task`` and ``readout_n_layers > 0`` on a small dense-coded graph: ``[C, 1]``."""
    _lib.require_device(x, f.w_last, rho.w_last, nam.w_last, g.code)
    return _SmallGraphNam.apply(x, g, bool(use_cnt), (f.L, f.H, f.F), (rho.L, rho.H), (nam.L, nam.H, nam.C),
                                *f[:6], *rho[:6], *nam[:6])


# =============================================================================
# one graph in slots: a batch-size-1 loop through ONE captured step
# =============================================================================
SLOT_MAX_NODES = 128
SLOT_CODES = 64              # most hop codes slots are laid out for (hops 0 .. 62 + the rest code); what gnan_small_batch_bwd covers
SLOT_CODE_TIERS = (16, 32, 64)   # ... and the capacities a loop picks from: the kernels' work on tables and bins grows with the capacity
_SLOT_CNT = None             # TensorKeyedCache: a graph's shell sizes (object identity + version) -> their [n, SLOT_CODES] layout


class SlotGraph:
    """ONE small dense-coded graph held in device slots whose SIZE only the device knows (``node_off = [0, n]``,
    ``code_off = [0, n * n]``): what a captured step of a graph-level task reads (trainer.py:23-86 feeds one graph of another
    size per step).  The batched kernels (``gnan_small_batch_fwd`` / ``_bwd``: blockIdx.y = graph, sizes from the offset
    arrays) run it as a batch of one, with the shell sizes (``cnt``, ABI 42) and the reference's rho arguments — so ONE capture
    serves every graph of up to 128 nodes and 63 hops, instead of one capture per (nodes, features, shells) shape.
    ``cnt`` is laid out for the slots' ``n_codes`` codes whatever the graph's own largest hop: its listed hops keep their columns, its
    rest bucket (unreachable pairs, code 255) moves to the last column, the columns between are hops no pair has."""
    is_dense = True

    def __init__(self, n_features: int, device, use_cnt: bool = True, max_nodes: int = SLOT_MAX_NODES, n_codes: int = SLOT_CODES):
        """``max_nodes``: 64 or 128 — the kernels come in a one-block (64 nodes, static LDS) and a two-block build; 97 % of
        Mutagenicity-shaped graphs fit the first, which is the faster one."""
        from .batched import HopBlocks
        if max_nodes not in (64, 128) or not 2 <= n_codes <= SLOT_CODES:
            raise ValueError("graph slots hold 64 or 128 nodes and 2 to 64 hop codes")
        self.F, self.device, self.n_codes, self.max_nodes = int(n_features), torch.device(device), int(n_codes), int(max_nodes)
        self.x = torch.zeros((max_nodes, self.F), dtype=torch.float32, device=device)
        self.code = torch.full((max_nodes * max_nodes,), 255, dtype=torch.uint8, device=device)
        self.cnt = torch.ones((max_nodes, self.n_codes), dtype=torch.int32, device=device) if use_cnt else None
        self.node_off = torch.zeros(2, dtype=torch.int32, device=device)
        self.code_off = torch.zeros(2, dtype=torch.int64, device=device)
        self.blocks = HopBlocks(self.code, self.node_off, self.code_off, [0], self.n_codes - 2)
        self.blocks.total_nodes, self.blocks.max_nodes, self.blocks.min_nodes, self.blocks.slots = max_nodes, max_nodes, 1, True
        self._offs = {}

    def fits(self, graph: HopGraph, x: torch.Tensor) -> bool:
        return bool(graph.is_dense and 1 <= graph.n_rows == graph.n_cols <= self.max_nodes and graph.n_codes <= self.n_codes
                    and x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (graph.n_rows, self.F)
                    and (self.cnt is None or graph.cnt is not None))

    def _cnt_layout(self, graph: HopGraph) -> torch.Tensor:
        global _SLOT_CNT
        if _SLOT_CNT is None:
            from ._cache import TensorKeyedCache
            _SLOT_CNT = TensorKeyedCache(1 << 16)
        hit = _SLOT_CNT.get((graph.cnt,), self.n_codes)
        if hit is None:
            D = graph.n_codes
            wide = torch.ones((graph.n_rows, self.n_codes), dtype=torch.int32, device=graph.cnt.device)
            wide[:, : D - 1] = graph.cnt[:, : D - 1]
            wide[:, self.n_codes - 1] = graph.cnt[:, D - 1]
            hit = _SLOT_CNT.put((graph.cnt,), self.n_codes, wide)
        return hit

    def load(self, graph: HopGraph, x: torch.Tensor, extra=()) -> None:
        """Copy a graph (that :meth:`fits`) into the slots — one launch (``gnan_multi_copy``); ``extra``: further (dst, src) pairs."""
        n = graph.n_rows
        offs = self._offs.get(n)
        if offs is None:
            offs = self._offs[n] = (torch.tensor([0, n], dtype=torch.int32, device=self.device),
                                    torch.tensor([0, n * n], dtype=torch.int64, device=self.device))
        pairs = [(self.x[:n], x), (self.code[: n * n], graph.code.reshape(-1)), (self.node_off, offs[0]), (self.code_off, offs[1])]
        if self.cnt is not None:
            pairs.append((self.cnt[:n], self._cnt_layout(graph)))
        pairs += list(extra)
        if all(sr.is_contiguous() and sr.dtype == d.dtype and sr.shape == d.shape and sr.is_cuda for d, sr in pairs) and len(pairs) <= 8:
            _lib.multi_copy(pairs)
            return
        for d, sr in pairs:
            d.copy_(sr)


def slot_mode(model):
    """``use_cnt`` (True / False) of the slot forward this model can run, or None: the read-out of a graph-level task with
    post-rho (or no) normalisation, a one-channel rho, features summed per node — ``models.TensorGNAN`` without a NAM read-out
    in its default order, the stand-alone class without normalisation.  Decided once per model."""
    mode = model.__dict__.get("_slot_mode", "?")
    if mode == "?":
        import types
        from . import modules
        mode = None
        if (isinstance(model, (modules.TensorGNAN, modules.StandaloneTensorGNAN)) and model.is_graph_task
                and not (isinstance(model, modules.TensorGNAN) and (model.readout_n_layers > 0 or model.aggregation_order == "reference"))
                and not (isinstance(model, modules.StandaloneTensorGNAN) and model.normalize_rho)):
            with torch.no_grad():
                f, rho = model._stacked("fs", model.fs), model._stacked("rho", [model.rho])
            if slot_graph_applies(types.SimpleNamespace(F=f.F), f, rho):
                mode = bool(model.normalize_rho)
        object.__setattr__(model, "_slot_mode", mode)
    return mode


def slot_tier(graph: HopGraph):
    """(node tier, hop-code tier) of the slots a dense-coded graph fits, or None."""
    if not graph.is_dense or graph.n_rows > SLOT_MAX_NODES or graph.n_codes > SLOT_CODES or graph.n_rows < 1:
        return None
    return (64 if graph.n_rows <= 64 else 128, next(c for c in SLOT_CODE_TIERS if graph.n_codes <= c))


def slot_graph_applies(slot: SlotGraph, f: StackedMLP, rho: StackedMLP) -> bool:
    return bool(f.F == slot.F and rho.F == 1 and f.L in (2, 3) and rho.L in (2, 3) and 1 <= f.H <= 64 and 1 <= rho.H <= 64
                and f.C <= 8 and rho.C == 1 and all(t is None or t.dtype == torch.float32 for t in tuple(f[:6]) + tuple(rho[:6])))


def slot_graph_forward(slot: SlotGraph, f: StackedMLP, rho: StackedMLP, use_cnt: bool) -> torch.Tensor:
    """The graph read-out ``[C, 1]`` (GNAN.py:75-79 / models.py:383-384) of the graph in the slots: one launch forward, two
    backward (``gnan_small_batch_bwd`` + the slab reduction)."""
    from .batched import _BatchedGraphs
    if use_cnt and slot.cnt is None:
        raise _lib.GnanHipError("these slots were laid out without shell sizes")
    fm, rm = (f.L, f.H, f.C, f.F), (rho.L, rho.H, rho.C)
    out = _BatchedGraphs.apply(slot.x, slot.blocks, True, fm, rm, slot.cnt if use_cnt else None, False, *f[:6], *rho[:6])   # [1, C]
    return out.reshape(-1, 1)
