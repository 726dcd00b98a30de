"""Node-level multi-GPU forward: 1-D vertex partition (edge-cut) with one all-gather per forward.

The reference is single-process (no torch.distributed anywhere); this is the build's own
scaling path for graphs of the RMAT / papers100M shapes.  One process per GPU; rank ``p`` owns the
contiguous node block ``[lo, hi)``: its rows of ``x``, its rows of the hop-coded CSR (global column ids)
and its rows of the output.

    step 1 (local)     operand_p = shape functions on owned rows            (HIP, no communication)
    step 2 (exchange)  operand   = all_gather(operand_p)                    (RCCL over xGMI)
    step 3 (local)     Y_p = rho-weighted aggregation over owned rows       (HIP)

The rest-bucket term needs the column sums of the full operand; every rank computes them from the
gathered operand, so no second collective is needed.  Parameters are replicated.

Feature partition (the cheaper exchange when the aggregation runs in the reference's order, W = F*C):
the operand COLUMNS are sharded instead — rank ``p`` owns features ``[k_lo, k_hi)`` for ALL nodes, computes its
slice of ``fx`` from its columns of ``x`` (no operand exchange at all), aggregates it over the whole graph
with the feature sum fused, and the ranks add their ``[N, C]`` partial outputs with one all-reduce.
xGMI is point-to-point (7 links per GPU): an all-gather of ``N*F*C`` floats is link-bound and costs more than
all the compute it feeds, while the all-reduce moves ``N*C`` floats — :func:`choose_partition` picks by bytes.

Halo recompute (what the bench uses for the reference order on more than one GPU): ``x`` is an input that does not
change between forwards, so rank ``p`` keeps, next to its own rows, a copy of the ``x`` rows of every remote node one
of its rows lists (its halo, found once per graph) and evaluates the shape functions for them itself.  Nothing of
size ``N`` crosses xGMI any more — the only collective is an all-reduce of the ``W`` column sums for the rest bucket
— at the price of ``n_halo * F`` redundant table look-ups per rank (1.4 ms per 10M x 64 on an MI355X, against
2-4 ms for the 320 MB shard every link would have to carry in the all-gather) and of HBM capacity, which is what a
288 GB device has to spare.

Halo exchange (``x`` must stay sharded: no rank may hold other ranks' input rows): the edge-cut scheme of SURVEY.md §8e
taken literally — rank ``p`` evaluates the shape functions of its OWN rows only and receives, from each owner, exactly the
operand rows its adjacency lists (deduplicated: its halo), as one all-to-all-v of point-to-point transfers; every
directed pair of ranks uses its own xGMI link.  Cheaper on the wire than gathering the whole operand (R-MAT 10M/100M at
8 ranks: 1.47M of 8.75M remote rows), dearer than recomputing the halo when ``x`` may be replicated.

Backward (every variant): each rank back-propagates the loss of ITS output rows and the ranks add their parameter
gradients (``all_reduce`` of ``p.grad``, SUM — the caller's job, as with any data-parallel step).  What crosses ranks
inside the backward pass mirrors the forward: the all-gather of the operand becomes a reduce-scatter of its gradient
(:class:`_GatherOperand`); the halo exchange runs in reverse and the owners add what comes back (:class:`_HaloExchange`); the all-reduced column sums of the rest bucket become an all-reduce of one W-float vector
(``rho_aggregate(total_group=...)``: every rank's rows pull on every rank's summed operand rows through ``total``);
the feature partition's all-reduce of the output passes the gradient through unchanged (every rank evaluates the same
loss on the full output).  All ranks must therefore run the backward pass together.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist


# A one-rank process group normally skips every collective.  With this switch on (``bench.py --force-dist``, the RCCL tests on a
# one-GPU box) a rank talks to its group whatever its size: communicator creation, every collective's RCCL launch and the
# all-reduce captured in :class:`SharePipeline`'s hipGraphs run on a world of one exactly as they would on eight.
ALWAYS_COMMUNICATE = False
# halo recompute, inference: look the owned rows up first and all-reduce their column sums under the halo rows' look-up
SPLIT_HALO_LOOKUP = True


def _communicates(world: int) -> bool:
    return world > 1 or (ALWAYS_COMMUNICATE and dist.is_available() and dist.is_initialized())


@dataclass
class VertexPartition:
    """Contiguous node blocks, one per rank.  ``bounds`` (``world + 1`` ascending node ids, ``bounds[0] = 0``,
    ``bounds[-1] = n_nodes``) cuts them anywhere — :func:`balanced_bounds` cuts by cost; without it the blocks have
    ``ceil(n_nodes / world)`` rows each (what the all-gather of :func:`gather_operand` needs: equally sized shards)."""
    n_nodes: int
    world: int
    rank: int
    bounds: Optional[tuple] = None

    def __post_init__(self):
        if self.bounds is not None:
            b = tuple(int(v) for v in self.bounds)
            if len(b) != self.world + 1 or b[0] != 0 or b[-1] != self.n_nodes or any(x > y for x, y in zip(b, b[1:])):
                raise ValueError(f"bounds must be {self.world + 1} ascending node ids from 0 to {self.n_nodes}, got {b}")
            self.bounds = b

    @property
    def uniform(self) -> bool:
        return self.bounds is None

    @property
    def block(self) -> int:                      # rows per rank, padded so every shard has the same size
        if self.bounds is not None:
            raise ValueError("a partition cut by cost has no common block size (the all-gather variant needs equal blocks)")
        return -(-self.n_nodes // self.world)

    @property
    def lo(self) -> int:
        if self.bounds is not None:
            return self.bounds[self.rank]
        return min(self.rank * self.block, self.n_nodes)

    @property
    def hi(self) -> int:
        if self.bounds is not None:
            return self.bounds[self.rank + 1]
        return min(self.lo + self.block, self.n_nodes)

    def owner_of(self, nodes: torch.Tensor) -> torch.Tensor:
        """Rank that owns each of ``nodes`` (int64 global ids)."""
        if self.bounds is None:
            return torch.div(nodes, self.block, rounding_mode="floor")
        edges = torch.tensor(self.bounds[1:-1], dtype=nodes.dtype, device=nodes.device)
        return torch.bucketize(nodes, edges, right=True)

    def lo_of(self, ranks: torch.Tensor) -> torch.Tensor:
        """First node of each of ``ranks``' blocks."""
        if self.bounds is None:
            return ranks * self.block
        return torch.tensor(self.bounds[:-1], dtype=ranks.dtype, device=ranks.device)[ranks]


# What a row of a share costs next to one stored pair (halo-recompute / exchange shares, reference order): the look-up of the
# row's F features against the aggregation of one pair's F-float operand row — 93 ps per 64-feature row against 40 ps per pair on
# an MI355X — and every stored pair of an R-MAT share brings ~0.11 halo rows to look up as well.
ROW_COST_IN_PAIRS = 2.3
HALO_ROWS_PER_PAIR = 0.11


def balanced_bounds(degree: torch.Tensor, world: int, row_cost: float = ROW_COST_IN_PAIRS,
                    halo_rows_per_pair: float = HALO_ROWS_PER_PAIR) -> tuple:
    """Cut ``[0, n)`` into ``world`` contiguous blocks of equal COST rather than equal row count: ``degree[i]`` stored pairs of
    row ``i`` (self pair included) at one unit each, plus ``row_cost`` per owned row and per halo row the pairs drag in.
    Equal row counts leave the slowest of eight R-MAT shares 2.4 % above the mean; this cut is within a few 1e-4 of it.
    Index arithmetic in float64 on whatever device ``degree`` lives; every rank computes the same cut from the same degrees."""
    n = int(degree.numel())
    if world <= 1 or n == 0:
        return (0, n) if world <= 1 else tuple([0] + [n] * world)
    cost = degree.to(torch.float64) * (1.0 + row_cost * halo_rows_per_pair) + row_cost
    cum = torch.cumsum(cost, 0)
    targets = cum[-1] * torch.arange(1, world, dtype=torch.float64, device=degree.device) / world
    cuts = (torch.searchsorted(cum, targets, right=False) + 1).clamp_(max=n).tolist()
    return tuple([0] + [int(c) for c in cuts] + [n])


def _hip_compute() -> Dict[str, Callable]:
    from .aggregate import add_rest_total_term, rest_total_term, rho_aggregate
    from .functional import column_sums, feature_mlps
    return {"feature_mlps": feature_mlps, "column_sums": column_sums, "aggregate": rho_aggregate,
            "rest_total_term": rest_total_term, "add_rest_total_term": add_rest_total_term}


def _reduce_scatter_sum(full: torch.Tensor, n_local: int, group) -> torch.Tensor:
    """Sum ``full [world * n_local, W]`` over the ranks and keep this rank's block.  RCCL: one reduce-scatter; any other
    backend (gloo in the CPU tests, mpi, ...): all-reduce and slice."""
    # chosen from the backend, on every rank alike — never by catching an error around a collective: a failure on one
    # rank only would leave the others inside a different collective
    if dist.get_backend(group) != "nccl":               # (gloo, mpi, ucc, custom backends: not every one has a reduce-scatter)
        full = full.clone()
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
        r = dist.get_rank(group)
        return full[r * n_local:(r + 1) * n_local].clone()
    out = full.new_empty((n_local, full.shape[1]))
    dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=group)
    return out


class _GatherOperand(torch.autograd.Function):
    """All-gather of equally sized row blocks; backward = reduce-scatter of the gradient."""

    @staticmethod
    def forward(ctx, local, group):
        ctx.group, ctx.n_local = group, local.shape[0]
        world = dist.get_world_size(group)
        full = local.new_empty((local.shape[0] * world, local.shape[1]))
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
        return full

    @staticmethod
    def backward(ctx, d_full):
        return _reduce_scatter_sum(d_full.contiguous(), ctx.n_local, ctx.group), None


class _SumOverRanks(torch.autograd.Function):
    """All-reduce (SUM) of partial outputs every rank then evaluates the SAME loss on: the gradient passes through."""

    @staticmethod
    def forward(ctx, partial, group):
        out = partial.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, d_out):
        return d_out, None


def gather_operand(local: torch.Tensor, part: VertexPartition, group=None) -> torch.Tensor:
    """All-gather the per-node operand rows; returns the ``[n_nodes, W]`` prefix of the padded buffer.  Differentiable:
    the gradient of the gathered operand is reduce-scattered back to the owners of its rows."""
    if not _communicates(part.world):
        return local
    W = local.shape[1]
    if local.shape[0] != part.block:
        local = torch.cat([local, local.new_zeros((part.block - local.shape[0], W))], dim=0)
    return _GatherOperand.apply(local, group)[: part.n_nodes]


def partitioned_forward(x_local: torch.Tensor, graph_local, stacked, lut: torch.Tensor, use_cnt: bool,
                        part: VertexPartition, order: str = "sum_first", out_channels: int = 1, group=None,
                        compute: Optional[Dict[str, Callable]] = None, marks: Optional[Callable] = None,
                        operand_dtype=torch.float32, tables=None):
    """Forward of the node-level path on this rank's block; returns ``out[lo:hi, :out_channels]``.

    ``order='sum_first'`` exchanges the narrow ``[N, C]`` operand (what the drop-in modules do);
    ``order='reference'`` exchanges ``[N, F*C]`` and sums over features after the aggregation
    (the evaluation order of models.py:373-376, which BASELINE's workload is stated in).
    ``compute`` lets the CPU/gloo tests substitute the kernels; the default is the HIP path.
    ``marks(name)`` is called between stages (bench.py records HIP events there).
    """
    ops = compute or _hip_compute()
    mark = marks or (lambda name: None)
    mark("start")
    # the column sums of the operand (rest-bucket total) come out of the shape-function pass; the ranks add
    # their W-float partials instead of re-reading the gathered operand
    kw = {} if operand_dtype == torch.float32 else {"out_dtype": operand_dtype}
    if compute is None and out_channels == 1:
        kw["pad_ok"] = True        # a ragged feature count may come back padded with zero columns: the read-out sums them away
    if tables is not None:
        kw["tables"] = tables
    operand_local, total = ops["feature_mlps"](x_local, stacked, order == "sum_first", return_total=True, **kw)
    mark("fmlp")
    operand = gather_operand(operand_local, part, group)
    if _communicates(part.world):            # one rank: no exchange, no event (every recorded event is a barrier packet)
        mark("gather")
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        mark("total")
    # reference order: the feature sum of models.py:375-376 rides in the aggregation kernel's epilogue.  `total` holds
    # the column sums of ALL gathered rows, so the backward pass needs no extra exchange for it: the rest-bucket vector
    # every rank adds to its gradient of the gathered operand is summed over the ranks by the reduce-scatter.
    Y = ops["aggregate"](graph_local, operand, lut, use_cnt, s_total=total,
                         reduce_channels=out_channels if order == "reference" else 0)
    mark("spmm")
    return Y


# =============================================================================
# feature partition
# =============================================================================
@dataclass
class FeaturePartition:
    n_features: int
    world: int
    rank: int

    @property
    def block(self) -> int:
        return -(-self.n_features // self.world)

    @property
    def lo(self) -> int:
        return min(self.rank * self.block, self.n_features)

    @property
    def hi(self) -> int:
        return min(self.lo + self.block, self.n_features)


def slice_features(stacked, lo: int, hi: int):
    """Rows ``[lo, hi)`` of every stacked per-feature tensor (the feature axis is dim 0, or dim 1 of ``*_mid``)."""
    def cut(t, axis):
        return None if t is None else t.narrow(axis, lo, hi - lo).contiguous()
    return type(stacked)(cut(stacked.w_first, 0), cut(stacked.b_first, 0), cut(stacked.w_mid, 1),
                         cut(stacked.b_mid, 1), cut(stacked.w_last, 0), cut(stacked.b_last, 0),
                         stacked.L, stacked.H, stacked.C, hi - lo)


def choose_partition(n_nodes: int, n_features: int, out_channels: int, world: int, order: str,
                     replicated_inputs: bool = True) -> str:
    """'vertex', 'feature' or 'halo', by the bytes each rank receives over xGMI per forward ('exchange' — the halo-only
    transfer of :func:`halo_exchange_forward` — is chosen explicitly: how many rows it moves depends on the graph).

    Sum-first exchanges the narrow ``[N, C]`` operand: vertex partition with one all-gather.  The reference order
    (``[N, F*C]`` operand) exchanges nothing if every rank may hold the ``x`` rows of its halo (halo recompute);
    if the inputs must stay sharded it is the feature partition unless the operand is narrower than two outputs."""
    if world == 1 or order == "sum_first":
        return "vertex"
    if replicated_inputs:
        return "halo"
    gather_bytes = n_nodes * n_features * out_channels * 4 * (world - 1) / world
    allreduce_bytes = 2 * n_nodes * out_channels * 4 * (world - 1) / world
    return "feature" if allreduce_bytes < gather_bytes else "vertex"


def feature_parallel_forward(x_cols: torch.Tensor, graph_full, stacked_local, lut: torch.Tensor, use_cnt: bool,
                             part: FeaturePartition, out_channels: int = 1, group=None,
                             compute: Optional[Dict[str, Callable]] = None, marks: Optional[Callable] = None,
                             operand_dtype=torch.float32):
    """Reference-order forward with the feature axis sharded; returns the full ``[N, out_channels]`` output
    (identical on every rank after the all-reduce).

    ``x_cols [N, k_hi-k_lo]`` are this rank's columns of ``x``; ``stacked_local`` the matching slice of the
    stacked shape-function weights (:func:`slice_features`); ``graph_full`` the whole hop-coded adjacency.
    """
    ops = compute or _hip_compute()
    mark = marks or (lambda name: None)
    mark("start")
    n = graph_full.n_rows
    if part.hi > part.lo:
        kw = {} if operand_dtype == torch.float32 else {"out_dtype": operand_dtype}
        operand, total = ops["feature_mlps"](x_cols, stacked_local, False, return_total=True, **kw)   # [N, Fp*C], [Fp*C]
        mark("fmlp")                                                           # no "gather" / "total" stage here
        Y = ops["aggregate"](graph_full, operand, lut, use_cnt, s_total=total, reduce_channels=out_channels)
    else:                                                                      # more ranks than features
        mark("fmlp")
        Y = x_cols.new_zeros((n, out_channels))
    mark("spmm")
    if _communicates(part.world):
        if torch.is_grad_enabled() and Y.requires_grad:
            Y = _SumOverRanks.apply(Y, group)
        else:
            dist.all_reduce(Y, op=dist.ReduceOp.SUM, group=group)
    mark("reduce")
    return Y


# =============================================================================
# halo recompute: vertex partition without an operand exchange
# =============================================================================
@dataclass
class HaloPlan:
    """Static per graph and rank: which operand rows the owned rows read and where they sit in the compact operand."""
    part: VertexPartition
    n_own: int
    halo: torch.Tensor            # int64 [n_halo]: global ids of the remote nodes listed by owned rows, ascending
    graph: object                 # HopGraph of the owned rows; column ids index the compact operand
                                  # (own node i -> i - lo, halo node -> n_own + its position in ``halo``)

    @property
    def n_needed(self) -> int:
        return self.n_own + int(self.halo.numel())

    def node_ids(self) -> torch.Tensor:
        """Global node id of every compact operand row (``x_compact = x[node_ids()]``)."""
        own = torch.arange(self.part.lo, self.part.hi, device=self.halo.device)
        return torch.cat([own, self.halo])


def build_halo_plan(graph_local, part: VertexPartition) -> HaloPlan:
    """``graph_local``: hop-coded CSR of the owned rows with GLOBAL column ids (e.g. ``synthetic.hop1_csr(..., lo, hi)``).
    Index work only (bit-exact, runs wherever the tensors live); the shell counts keep referring to the whole graph."""
    from .graph import HopGraph
    lo, hi = part.lo, part.hi
    n_own = hi - lo
    col = graph_local.col.long()
    own = (col >= lo) & (col < hi)
    halo = torch.unique(col[~own])                                   # sorted
    pos = torch.searchsorted(halo, col) if halo.numel() else torch.zeros_like(col)
    compact = torch.where(own, col - lo, n_own + pos)
    g = HopGraph.from_csr(graph_local.rowptr, compact.to(torch.int32), graph_local.code,
                          n_cols=n_own + int(halo.numel()), n_codes=graph_local.n_codes, cnt=graph_local.cnt)
    return HaloPlan(part, n_own, halo, g)


def halo_recompute_forward(x_compact: torch.Tensor, plan: HaloPlan, stacked, lut: torch.Tensor, use_cnt: bool,
                           order: str = "reference", out_channels: int = 1, group=None,
                           compute: Optional[Dict[str, Callable]] = None, marks: Optional[Callable] = None,
                           operand_dtype=torch.float32, tables=None):
    """Forward on the owned rows from ``x_compact = x[plan.node_ids()]``; returns ``out[lo:hi, :out_channels]``.
    Same stages and ``marks`` as :func:`partitioned_forward`; there is no "gather" stage, and "total" only with > 1 rank."""
    ops = compute or _hip_compute()
    mark = marks or (lambda name: None)
    part = plan.part
    mark("start")
    kw = {} if operand_dtype == torch.float32 else {"out_dtype": operand_dtype}
    if compute is None and out_channels == 1:
        kw["pad_ok"] = True        # a ragged feature count may come back padded with zero columns: the read-out sums them away
    if tables is not None:
        kw["tables"] = tables      # built ahead of time on a side stream (functional.TablePrefetch; inference loops)
    sum_first = order == "sum_first"
    rc = out_channels if order == "reference" else 0
    if (SPLIT_HALO_LOOKUP and compute is None and _communicates(part.world) and not torch.is_grad_enabled() and stacked is not None
            and stacked.F % 16 == 0 and 0 < plan.n_own < plan.n_needed):
        # inference: the OWNED rows' look-up first — the column sums are theirs alone — then the all-reduce of those 4*W bytes
        # runs under the halo rows' look-up (more than half of a share's look-up work on an R-MAT graph) and has landed when the
        # aggregation starts: it takes the real sums, one launch, no rest-term pass behind it.  Both look-ups write into one
        # operand; the row views are kept per input matrix (the look-up's range hint is remembered per tensor object).
        views = getattr(plan, "_row_views", None)
        if views is None or views[0] is not x_compact:
            views = plan._row_views = (x_compact, x_compact[: plan.n_own], x_compact[plan.n_own:])
        width = stacked.C if sum_first else stacked.F * stacked.C
        operand = torch.empty((plan.n_needed, width), dtype=operand_dtype, device=x_compact.device)
        rows_own, rows_halo = operand[: plan.n_own], operand[plan.n_own:]
        own, total = ops["feature_mlps"](views[1], stacked, sum_first, return_total=True, out=rows_own, **kw)
        if own is not rows_own:                               # (a look-up that could not take the caller's rows)
            rows_own.copy_(own)
        work = dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group, async_op=True)
        halo = ops["feature_mlps"](views[2], stacked, sum_first, out=rows_halo, **kw)
        if halo is not rows_halo:
            rows_halo.copy_(halo)
        mark("fmlp")
        work.wait()
        mark("total")
        Y = ops["aggregate"](plan.graph, operand, lut, use_cnt, s_total=total, reduce_channels=rc, total_rows=plan.n_own)
        mark("spmm")
        return Y
    # the column sums ride in the shape-function pass, restricted to the owned rows: they partition the nodes, so the
    # ranks' sums add up to the whole graph's without double counting
    operand, total = ops["feature_mlps"](x_compact, stacked, sum_first, return_total=True, total_rows=plan.n_own, **kw)
    mark("fmlp")
    if _communicates(part.world) and not torch.is_grad_enabled() and "rest_total_term" in ops:
        # inference: the aggregation does not wait for the all-reduce of the 256-byte column sums (tens of microseconds of
        # collective latency against ~0.6 ms of kernel on a 1/8 share) — it runs against zero sums, i.e. computes
        # sum_d (wt_d - wt_rest) * S, and the wt_rest * total term is added once the collective has landed
        work = dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group, async_op=True)
        mark("total")
        Y = ops["aggregate"](plan.graph, operand, lut, use_cnt, s_total=torch.zeros_like(total), reduce_channels=rc)
        work.wait()
        if "add_rest_total_term" in ops:                  # one launch, in place
            ops["add_rest_total_term"](Y, plan.graph, lut, use_cnt, total, rc)
        else:
            Y += ops["rest_total_term"](plan.graph, lut, use_cnt, total, rc)
        mark("spmm")
        return Y
    shared = {}
    if _communicates(part.world):
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        mark("total")
        shared = {"total_group": group}          # backward: the ranks add their rest-bucket vectors over the same group
    # only the owned rows went into `total` (halo rows are some other rank's owned rows)
    Y = ops["aggregate"](plan.graph, operand, lut, use_cnt, s_total=total, reduce_channels=rc, total_rows=plan.n_own,
                         **shared)
    mark("spmm")
    return Y


# =============================================================================
# halo exchange: vertex partition, only the listed remote operand rows travel
# =============================================================================
@dataclass
class ExchangePlan:
    """Who sends which of its owned operand rows to whom (static per graph and partition)."""
    halo: HaloPlan
    send_rows: list               # per peer rank: int64 local indices (into the owned rows) this rank sends, in the peer's halo order
    recv_counts: list             # per peer rank: how many halo rows it owns (the halo is sorted by global id = grouped by owner)

    @property
    def n_own(self) -> int:
        return self.halo.n_own


def _peer_exchange(send: list, recv: list, group=None) -> None:
    """All-to-all-v as a batch of point-to-point transfers (RCCL groups them; gloo, which has no all_to_all, takes them too)."""
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "gloo" and any(t is not None and t.is_cuda for t in list(send) + list(recv)):
        # gloo moves point-to-point messages through host memory only (the 1-GPU dry run of the multi-rank bench)
        host_send = [None if t is None else t.cpu() for t in send]
        host_recv = [None if t is None else torch.empty(t.shape, dtype=t.dtype) for t in recv]
        _peer_exchange(host_send, host_recv, group)
        for dst, src in zip(recv, host_recv):
            if dst is not None and dst.numel():
                dst.copy_(src)
        return
    ops = []
    for r, (s_buf, r_buf) in enumerate(zip(send, recv)):
        if r == rank:
            continue
        peer = dist.get_global_rank(group, r) if group is not None else r
        if r_buf is not None and r_buf.numel():
            ops.append(dist.P2POp(dist.irecv, r_buf, peer, group))
        if s_buf is not None and s_buf.numel():
            ops.append(dist.P2POp(dist.isend, s_buf, peer, group))
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()


def build_exchange_plan(graph_local, part: VertexPartition, group=None) -> ExchangePlan:
    """Halo plan (:func:`build_halo_plan`) plus the send lists: every rank tells the owners of its halo rows which rows it
    needs (one exchange of counts, one of index lists — once per graph).  Index work only, bit-exact."""
    plan = build_halo_plan(graph_local, part)
    world, dev = part.world, plan.halo.device
    owner = part.owner_of(plan.halo)
    recv_counts = torch.bincount(owner, minlength=world).tolist()
    if not _communicates(world):
        return ExchangePlan(plan, [None], recv_counts)
    counts = torch.tensor(recv_counts, dtype=torch.int64, device=dev)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)              # all_counts[q][r]: rows rank q needs from rank r
    want = list(torch.split(plan.halo - part.lo_of(owner), recv_counts))     # per owner: ITS local indices, in my halo order
    asked = [torch.empty(int(all_counts[q][part.rank]), dtype=torch.int64, device=dev) for q in range(world)]
    _peer_exchange([w.contiguous() for w in want], asked, group)
    return ExchangePlan(plan, asked, recv_counts)


class _HaloExchange(torch.autograd.Function):
    """``halo_rows = exchange(owned_rows)``: each peer receives the owned rows it listed; backward sends the gradients of
    the halo rows back to their owners, which add them to their rows."""

    @staticmethod
    def forward(ctx, own, xplan: ExchangePlan, group):
        ctx.xplan, ctx.group, ctx.n_own = xplan, group, own.shape[0]
        W = own.shape[1]
        send = [None if idx is None else own.index_select(0, idx) for idx in xplan.send_rows]
        halo = own.new_empty((sum(xplan.recv_counts), W))
        recv = list(torch.split(halo, xplan.recv_counts))
        _peer_exchange(send, recv, group)
        return halo

    @staticmethod
    def backward(ctx, d_halo):
        xplan = ctx.xplan
        d_halo = d_halo.contiguous()
        send = list(torch.split(d_halo, xplan.recv_counts))
        recv = [None if idx is None else d_halo.new_empty((idx.numel(), d_halo.shape[1])) for idx in xplan.send_rows]
        _peer_exchange(send, recv, ctx.group)
        d_own = d_halo.new_zeros((ctx.n_own, d_halo.shape[1]))
        for idx, buf in zip(xplan.send_rows, recv):
            if idx is not None and idx.numel():
                d_own.index_add_(0, idx, buf)
        return d_own, None, None


def halo_exchange_forward(x_own: torch.Tensor, xplan: ExchangePlan, stacked, lut: torch.Tensor, use_cnt: bool,
                          order: str = "sum_first", out_channels: int = 1, group=None,
                          compute: Optional[Dict[str, Callable]] = None, marks: Optional[Callable] = None,
                          operand_dtype=torch.float32):
    """Forward on the owned rows from the OWNED rows of ``x`` only; returns ``out[lo:hi, :out_channels]``.
    Stages / ``marks``: "fmlp" (own shape functions), "gather" (the halo rows arrive), "total", "spmm"."""
    ops = compute or _hip_compute()
    mark = marks or (lambda name: None)
    plan = xplan.halo
    part = plan.part
    mark("start")
    kw = {} if operand_dtype == torch.float32 else {"out_dtype": operand_dtype}
    if compute is None and out_channels == 1:
        kw["pad_ok"] = True
    own, total = ops["feature_mlps"](x_own, stacked, order == "sum_first", return_total=True, **kw)
    mark("fmlp")
    shared = {}
    if _communicates(part.world):
        operand = torch.cat([own, _HaloExchange.apply(own, xplan, group)], dim=0)       # [own | halo]: the plan's column order
        mark("gather")
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        mark("total")
        shared = {"total_group": group}
    else:
        operand = own
    Y = ops["aggregate"](plan.graph, operand, lut, use_cnt, s_total=total,
                         reduce_channels=out_channels if order == "reference" else 0, total_rows=plan.n_own, **shared)
    mark("spmm")
    return Y


# =============================================================================
# a rank's inference loop without the host in it
# =============================================================================
class SharePipeline:
    """Inference loop of one rank's share, replayed from hipGraphs: ``step()`` is ONE graph launch.

    A share of the 10M-node graph on 8 ranks is ~0.9 ms of kernels behind ~12 launches; issued from Python the loop is
    bound by the host.  Two captured graphs alternate.  Graph ``p`` holds the whole forward — look-up from table set ``p``,
    the all-reduce of the column sums (captured with the RCCL backend; the aggregation does not wait for it), aggregation —
    and, on a forked branch that runs under them, everything the NEXT forward needs built from the weights: its tables
    into set ``1 - p`` (``functional.TablePrefetch``), their direct-index tables and the check that they fit the captured
    look-up.  Every forward still consumes a build of its own, made from the weights as they are while the previous
    forward runs — an inference loop.  :meth:`tripped` reads the guard the checks set.

    ``forward(tables, marks)`` is the share's forward given pre-built tables (e.g. a ``halo_recompute_forward`` closure
    passing both on).  ``fork_at``: the stage mark at which the branch starts ("start": under the look-up, "fmlp": under
    the aggregation).  ``x``: the feature matrix the look-up reads (its value range positions the direct-index grid).
    Raises ``graphed.CaptureFailed`` when the step cannot be captured (a backend whose collectives need the host, e.g.
    gloo); the caller then keeps its eager loop."""

    def __init__(self, forward: Callable, stacked, x: Optional[torch.Tensor] = None, warmup: int = 2, fork_at: str = "start"):
        from . import functional, pwl
        from .graphed import CaptureFailed, GraphedCallable
        self.prefetch = functional.TablePrefetch(stacked)
        if not self.prefetch.applies:
            raise CaptureFailed("the table build of these shape functions is not a kernel (L > 3 or H > 128)")
        st = self.prefetch.stacked
        dev = st.w_last.device
        self.bufs = [pwl.table_buffers(st) for _ in range(2)]
        self.guard = torch.zeros(1, dtype=torch.float32, device=dev)
        x_range = functional._feature_range(x) if (x is not None and st.C == 1) else None
        self.index = [None, None]
        if x_range is not None:
            self.index = [(torch.empty((st.F, functional.INDEX_BUCKETS), dtype=torch.int16, device=dev),
                           torch.empty((st.F, 2), dtype=torch.float32, device=dev)) for _ in range(2)]
        main = torch.cuda.current_stream(dev)

        def launch(slot, guard=None):
            return self.prefetch.launch(buffers=self.bufs[slot], x_range=x_range, index_buffers=self.index[slot], guard=guard)

        first = launch(0)
        main.wait_stream(self.prefetch.side)
        with torch.no_grad():
            forward(first, None)                            # eager: table sizes become known (the speculative plan)
        # every build the captured look-ups consume is checked against the captured sizes on the SIDE stream, by the replay
        # that made it (launch(..., guard)); the eager builds of the warm-ups likewise, so that no check is captured on the
        # main stream between look-up and aggregation (7 us of a 0.9-ms share)
        self.pending = [launch(0, self.guard), None]
        main.wait_stream(self.prefetch.side)
        self.graphs = []
        for p in (0, 1):
            if p == 1:
                # graph 1 looks up from set 1: an eager build of it for the warm-up forwards (what graph 0's CAPTURE left
                # in pending[1] describes a build that has not run)
                self.pending[1] = launch(1, self.guard)
                main.wait_stream(self.prefetch.side)

            def fn(p=p):
                made = []

                def marks(name):
                    if name == fork_at and not made:
                        made.append(launch(1 - p, self.guard))          # forked branch: the next forward's tables
                with torch.no_grad():
                    out = forward(self.pending[p], marks)
                if not made:
                    made.append(launch(1 - p, self.guard))
                torch.cuda.current_stream(dev).wait_stream(self.prefetch.side)   # join
                self.pending[1 - p] = made[0]
                return out
            self.graphs.append(GraphedCallable(fn, warmup=warmup, guard=self.guard))
        self.parity = 0          # set 0 holds a build of the current weights: graph 1's eager warm-ups wrote it last
        self.guard.zero_()

    def step(self) -> torch.Tensor:
        """One forward; the returned tensor is overwritten by the next-but-one call."""
        out = self.graphs[self.parity].replay()
        self.parity ^= 1
        return out

    def tripped(self) -> bool:
        """Did any replay meet tables that outgrew the captured look-up (its output is then not to be trusted)?"""
        return bool(self.guard.item() != 0)
