"""Exact piecewise-linear tables of the per-feature shape functions.

Each ``f_k : R -> R^C`` is a ReLU MLP of a SCALAR input (GNAN.py:24-34), hence exactly piecewise linear in
``x``: its kinks are the points where some hidden unit's pre-activation crosses zero.  For the reference's
default depth the count is small (H kinks from the first layer plus a handful per later unit — ~130 for
H = 64, L = 3), so ``f_k`` can be tabulated *exactly*, once per forward, from the current weights:

    f_k(x) = val[i] + slope[i] * (x - anchor[i]),      i = #{ breakpoints of f_k <= x }

and evaluating N x F shape functions becomes N x F table look-ups (``gnan_fpwl_fwd``) instead of
2·N·F·H² flops.  This is the same idea the aggregation uses for rho (evaluate the network only where its
argument can actually change behaviour), applied to ``f``.

The tables are built here with a few batched float64 torch ops on the device (O(F · P · H²) flops, P ~ 10²);
kink positions are found layer by layer from sign changes of the pre-activations between consecutive
breakpoints (each pre-activation is affine there), and the table values are the network itself evaluated in
float64 at the float32-rounded breakpoints — so the tabulated function agrees with the float64 reference to
float32 round-off, independent of N.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch


MAX_PIECES = 4096          # per feature; beyond this the tabulation is refused (use the matrix-core kernel)


LDS_PREFERRED = 48 * 1024  # table bytes per feature group that still leaves >= 3 workgroups per CU
LDS_LIMIT = 150 * 1024


class PwlTables(NamedTuple):
    off: torch.Tensor      # int32 [F + 1]   piece offsets; feature k owns pieces off[k] .. off[k+1]-1
    anchor: torch.Tensor   # fp32  [T]       anchor[off[k] + i] = left breakpoint of piece i (piece 0: the first breakpoint)
    val: torch.Tensor      # fp32  [T, C]    f_k(anchor)
    slope: torch.Tensor    # fp32  [T, C]
    max_pieces: int
    features_per_group: int   # consecutive features sharing one LDS image in gnan_fpwl_fwd
    max_group_pieces: int


def table_stride(C: int) -> int:
    """Floats per (val | slope) row of the LDS tables: odd for C > 1 (bank conflicts, csrc/fpwl.hip ``table_stride``)."""
    return (C | 1) if C > 1 else 1


def _plan_groups(off, C):
    """Largest power-of-two group (<= 16 features) whose tables fit the preferred LDS budget; if none does (many
    channels), the largest group whose 64-bit moment bins ALSO fit LDS — the float-bin fallback of the backward pass is an
    order of magnitude slower (LDS float atomics are a compare-and-swap loop on gfx950) —, else the largest that fits."""
    F = len(off) - 1
    fits = []
    for fg in (16, 8, 4, 2, 1):
        mg = max(off[min(F, k + fg)] - off[k] for k in range(0, F, fg))
        nbytes = mg * (1 + 2 * table_stride(C)) * 4
        if nbytes <= LDS_PREFERRED:
            return fg, mg
        if nbytes <= LDS_LIMIT:
            bins = (mg + 1) // 2 * 8 + mg * (2 * C + 1 if C > 1 else 2) * 8
            fits.append((fg, mg, bins <= LDS_LIMIT))
    for fg, mg, bins_fit in fits:
        if bins_fit:
            return fg, mg
    if fits:
        return fits[0][:2]
    # not even one feature's tables fit LDS (C > ~110 channels): the thread-per-node kernels cannot run, the two-phase
    # kernels (csrc/fpwl_rows.hip: tables stay in global memory) can — functional._fpwl_rows_applies sends such tables there
    return (1, max(off[k + 1] - off[k] for k in range(F))) if C > 1 else None


def oversize(t: "PwlTables") -> bool:
    """Tables whose largest feature group does not fit the LDS image of the thread-per-node kernels."""
    C = t.val.shape[1]
    return t.max_group_pieces * (1 + 2 * table_stride(C)) * 4 > LDS_LIMIT


def _prefix(x: torch.Tensor, p, depth: int, w, b):
    """Pre-activations of hidden layer ``depth`` (0-based) at points ``x [F, M]`` -> ``[F, M, H]`` (float64)."""
    z = x.unsqueeze(-1) * w[0].unsqueeze(1)
    if b[0] is not None:
        z = z + b[0].unsqueeze(1)
    for l in range(1, depth + 1):
        z = torch.bmm(torch.relu(z), w[l].transpose(1, 2))
        if b[l] is not None:
            z = z + b[l].unsqueeze(1)
    return z


def _network(x: torch.Tensor, p, w, b, w_last, b_last):
    """Full ``f_k`` at ``x [F, M]`` -> ``[F, M, C]`` in float64."""
    if p.L == 1:
        y = x.unsqueeze(-1) * w_last.unsqueeze(1)
    else:
        h = torch.relu(_prefix(x, p, p.L - 2, w, b))
        y = torch.bmm(h, w_last.transpose(1, 2))
    if b_last is not None:
        y = y + b_last.unsqueeze(1)
    return y


def _round_up_f32(bp: torch.Tensor) -> torch.Tensor:
    """Float64 kinks -> the smallest float32 not below them (+inf padding stays).  A node is sent to piece
    ``#{anchors <= x}``; with anchors rounded UP, ``x >= anchor`` implies ``x >=`` the true kink, so a float32 ``x`` that
    equals an anchor is never put on the wrong side of a kink that lies strictly between two float32 numbers."""
    c = torch.where(torch.isfinite(bp), bp.clamp(-3.0e38, 3.0e38), bp)
    f = c.float()
    below = torch.isfinite(f) & (f.double() < c)
    return torch.where(below, torch.nextafter(f, torch.full_like(f, float("inf"))), f)


def _with_point_pieces(bp32: torch.Tensor, p, w, b) -> torch.Tensor:
    """Anchors ``[F, P]`` (sorted, +inf padded) -> ``[F, 2P]`` with ``nextafter(a)`` inserted behind every anchor ``a``
    at which some hidden unit's pre-activation is EXACTLY zero (zero biases put every first-layer kink at x = 0,
    GNAN.py:49-53; one-hot and bag-of-words features are mostly exact zeros).  The piece ``[a, nextafter(a))`` then
    holds the nodes with ``x == a`` and nothing else, and the backward pass differentiates it AT ``a`` — where torch
    takes relu'(0) = 0 for the unit that sits on its kink — instead of inside the piece to the right
    (:func:`piece_probe_points`, ``csrc/fpwl_grad.hip:piece_points``).  The tabulated function is unchanged."""
    F, P = bp32.shape
    INF = float("inf")
    finite = torch.isfinite(bp32)
    t = torch.where(finite, bp32, torch.zeros_like(bp32)).double()
    on_kink = torch.zeros_like(finite)
    for depth in range(p.L - 1):
        hit = _prefix(t, p, depth, w, b) == 0                        # [F, P, H]
        if depth == 0:
            hit = hit & (w[0] != 0).unsqueeze(1)                     # w = 0: a constant unit has no kink
        on_kink = on_kink | hit.any(dim=-1)
    up = torch.nextafter(bp32, torch.full_like(bp32, INF))
    nxt = torch.cat([bp32[:, 1:], torch.full_like(bp32[:, :1], INF)], dim=1)
    # only behind the last of a run of coinciding anchors, and only where the next anchor does not already end the piece there
    extra = torch.where(on_kink & finite & (nxt > up), up, torch.full_like(up, INF))
    return torch.sort(torch.cat([bp32, extra], dim=1), dim=1)[0]


def _build_padded(p):
    """Device part of the build — static shapes only, no host round trip (so it can live in a hipGraph):
    returns ``(off [F+1] int64, overflow [] bool, anchor [F, P+1], val [F, P+1, C], slope [F, P+1, C], keep [F, P+1])``
    where ``keep`` marks the first ``pieces_k`` slots of every feature."""
    dev = p.w_last.device
    f64 = torch.float64
    F = p.F
    INF = float("inf")
    if p.L >= 2:
        w = [p.w_first.to(f64)] + [p.w_mid[l].to(f64) for l in range(p.L - 2)]
        b = [None if p.b_first is None else p.b_first.to(f64)] + \
            [None if p.b_mid is None else p.b_mid[l].to(f64) for l in range(p.L - 2)]
    else:
        w, b = [], []
    w_last = p.w_last.to(f64)
    b_last = None if p.b_last is None else p.b_last.to(f64)

    bp = torch.full((F, 0), INF, dtype=f64, device=dev)             # sorted breakpoints, +inf padded
    cap_per_layer = max(64, 4 * max(p.H, 1))                        # ~2H kinks per layer in practice (see module doc)
    overflow = torch.zeros((), dtype=torch.bool, device=dev)
    for depth in range(p.L - 1):                                     # every hidden layer adds its units' zero crossings
        if depth == 0:
            b0 = b[0] if b[0] is not None else torch.zeros_like(w[0])
            new = torch.where(w[0] != 0, -b0 / w[0], torch.full_like(w[0], INF))       # [F, H]
        else:
            finite = torch.isfinite(bp)
            any_f = finite.any(dim=1, keepdim=True)
            t_first = torch.where(any_f, bp[:, :1], torch.zeros_like(bp[:, :1]))
            t_last = torch.where(any_f, torch.where(finite, bp, torch.full_like(bp, -INF)).amax(dim=1, keepdim=True),
                                 torch.zeros_like(bp[:, :1]))
            pad = torch.where(finite, bp, t_last.expand_as(bp))
            nodes = torch.cat([t_first - 2, t_first - 1, pad, t_last + 1, t_last + 2], dim=1)     # [F, P + 4]
            z = _prefix(nodes, p, depth, w, b)                                                    # [F, P + 4, H]
            e0, e1 = nodes[:, :-1].unsqueeze(-1), nodes[:, 1:].unsqueeze(-1)
            z0, z1 = z[:, :-1], z[:, 1:]
            inside = (z0 * z1) < 0                                   # sign change strictly inside (e0, e1)
            root = e0 + (e1 - e0) * (z0 / (z0 - z1))
            new_in = torch.where(inside, root, torch.full_like(root, INF))
            # rays: extrapolate the affine piece beyond the outermost sample points
            dl = z[:, 1] - z[:, 0]
            rl = nodes[:, :1] - z[:, 0] / dl * (nodes[:, 1:2] - nodes[:, :1])
            ok_l = (dl != 0) & (z[:, 0] / dl > 0)
            dr = z[:, -1] - z[:, -2]
            rr = nodes[:, -1:] - z[:, -1] / dr * (nodes[:, -1:] - nodes[:, -2:-1])
            ok_r = (dr != 0) & (z[:, -1] / dr < 0)
            new = torch.cat([new_in.reshape(F, -1), torch.where(ok_l, rl, torch.full_like(rl, INF)),
                             torch.where(ok_r, rr, torch.full_like(rr, INF))], dim=1)
        new = torch.where(torch.isfinite(new), new, torch.full_like(new, INF))       # NaN / -inf -> dropped
        bp, _ = torch.sort(torch.cat([bp, new], dim=1), dim=1)
        # keep a fixed number of columns per layer (no host round trip); an overflow is detected at the end
        cap = min(bp.shape[1], cap_per_layer * (depth + 1))
        if bp.shape[1] > cap:
            overflow = overflow | torch.isfinite(bp[:, cap]).any()
            bp = bp[:, :cap]
    if bp.shape[1] == 0:                                             # L == 1: affine, no kinks
        bp = torch.full((F, 1), INF, dtype=f64, device=dev)

    # float32 anchors (what the kernel compares x against), network values there in float64
    bp32 = _with_point_pieces(_round_up_f32(bp), p, w, b)
    finite = torch.isfinite(bp32)
    n_bp = finite.sum(dim=1)                                          # [F]
    P = bp32.shape[1]
    t = bp32.to(f64)
    any_f = (n_bp > 0).unsqueeze(1)
    t_first = torch.where(any_f, t[:, :1], torch.zeros((F, 1), dtype=f64, device=dev))
    t_last = torch.where(any_f, torch.where(finite, t, torch.full_like(t, -INF)).amax(dim=1, keepdim=True),
                         torch.zeros_like(t_first))
    tp = torch.where(finite, t, t_last.expand_as(t))
    nodes = torch.cat([t_first - 1, tp, t_last + 1], dim=1)          # [F, P + 2]
    v = _network(nodes, p, w, b, w_last, b_last)                     # [F, P + 2, C]
    width = (nodes[:, 1:] - nodes[:, :-1]).unsqueeze(-1)             # [F, P + 1, 1]
    sl = torch.where(width > 0, (v[:, 1:] - v[:, :-1]) / width, torch.zeros_like(v[:, 1:]))   # slope of piece i
    # piece i (0..P): anchor = its left breakpoint, except piece 0 which is anchored at the first breakpoint
    anchor = torch.cat([nodes[:, 1:2], nodes[:, 1:-1]], dim=1)       # [F, P + 1]
    val = torch.cat([v[:, 1:2], v[:, 1:-1]], dim=1)                  # [F, P + 1, C]
    # features with fewer breakpoints: their trailing pieces are zero-width copies; keep only n_bp + 1 pieces
    pieces = n_bp + 1
    # the right ray's slope lives at index n_bp of `sl` only when n_bp == P; gather it per feature
    idx = torch.arange(P + 1, device=dev).unsqueeze(0)
    ray = sl[:, -1:].expand(-1, P + 1, -1)
    sl = torch.where((idx == n_bp.unsqueeze(1)).unsqueeze(-1), ray, sl)
    keep = idx < pieces.unsqueeze(1)
    off = torch.zeros(F + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(pieces, 0)
    return torch.cat([off, overflow.to(torch.int64).view(1)]), anchor.float(), val.float(), sl.float(), keep


class _GraphedBuild:
    """The ~60 tiny launches of ``_build_padded`` captured once per model shape into a hipGraph and replayed.

    Per forward: one fused copy of the current weights into the graph's static inputs, one graph launch.
    Falls back to eager execution if capture is not possible (CPU tensors, capture already in progress, ...)."""

    _cache = {}

    def __init__(self, p):
        live = [t for t in p[:6] if t is not None]
        self.static_in = [torch.empty_like(t) for t in live]
        self.present = [t is not None for t in p[:6]]
        torch._foreach_copy_(self.static_in, live)
        it = iter(self.static_in)
        sp = type(p)(*[next(it) if pr else None for pr in self.present], *p[6:])
        side = torch.cuda.Stream(device=p.w_last.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                       # warm-up outside capture (lazy inits, allocator)
                _build_padded(sp)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = _build_padded(sp)

    def __call__(self, p):
        torch._foreach_copy_(self.static_in, [t for t in p[:6] if t is not None])
        self.graph.replay()
        return self.static_out

    @classmethod
    def run(cls, p):
        dev = p.w_last.device
        if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return _build_padded(p)
        key = (dev, p.F, p.L, p.H, p.C, tuple(t is not None for t in p[:6]))
        g = cls._cache.get(key)
        if g is None:
            try:
                g = cls(p)
            except Exception:                        # capture unsupported here: remember and run eagerly
                g = False
            cls._cache[key] = g
        return g(p) if g else _build_padded(p)


_PINNED_META = {}


def _pinned_meta(dev, n: int, slot: int = 0) -> torch.Tensor:
    key = (dev, n, slot)
    buf = _PINNED_META.get(key)
    if buf is None:
        buf = _PINNED_META[key] = torch.empty(n, dtype=torch.int32, pin_memory=True)
    return buf


def table_buffers(p):
    """Caller-owned output buffers of one table build (``_build_tables_hip(buffers=...)``): a loop that replays captured
    look-ups needs its tables at fixed addresses (``distributed.SharePipeline`` alternates between two sets)."""
    from . import _lib
    dev = p.w_last.device
    F, C = p.F, p.C
    cap = min(1024, max(64, 4 * p.H) * (p.L - 1))
    T = F * (cap + 1)
    need = _lib.lib().gnan_pwl_build_scratch_bytes(F, C, cap)
    return (torch.empty(T, dtype=torch.float32, device=dev), torch.empty((T, C), dtype=torch.float32, device=dev),
            torch.empty((T, C), dtype=torch.float32, device=dev), torch.empty(F + 2, dtype=torch.int32, device=dev),
            torch.empty(need // 8 + 1, dtype=torch.float64, device=dev))


INDEX_IN_BUILD = True        # the look-up's direct-index tables out of the build's compaction pass (gnan_pwl_build_args.index_*)


def _build_tables_hip(p, lazy: bool = False, pinned_slot: int = 0, buffers=None, index_request=None):
    """Tables by TWO kernel launches (``gnan_pwl_build``: a workgroup per feature finds the kinks and tabulates the
    network in float64, LDS-resident; a second tiny kernel packs the features back to back) and one
    device->host copy of the F+1 offsets.  Covers L in {2, 3}, H <= 128; same result as :func:`_build_padded`."""
    from . import _lib
    dev = p.w_last.device
    F, C = p.F, p.C
    cap = min(1024, max(64, 4 * p.H) * (p.L - 1))
    if buffers is None:
        buffers = table_buffers(p)                                    # (off[F+1] | overflow: every entry is written by the build)
    anchor, val, slope, meta, scratch = buffers
    keepalive = [t if t is None else t.detach().float().contiguous() for t in p[:6]]
    w_mid = keepalive[2][0] if keepalive[2] is not None else None      # [1, F, H, H] -> [F, H, H]
    b_mid = keepalive[3][0] if keepalive[3] is not None else None
    a = _lib.PwlBuildArgs(w_first=_lib.ptr(keepalive[0]), b_first=_lib.ptr(keepalive[1]), w_mid=_lib.ptr(w_mid),
                          b_mid=_lib.ptr(b_mid), w_last=_lib.ptr(keepalive[4]), b_last=_lib.ptr(keepalive[5]),
                          F=F, L=p.L, H=p.H, C=C, cap=cap, anchor=_lib.ptr(anchor), val=_lib.ptr(val),
                          slope=_lib.ptr(slope), off=_lib.ptr(meta), overflow=meta[F + 1:].data_ptr(),
                          scratch=_lib.ptr(scratch), scratch_bytes=scratch.numel() * 8)
    index = None
    if index_request is not None and INDEX_IN_BUILD:
        # (x_range [F, 2], buckets[, (table, key) caller-owned]): one launch less than gnan_fpwl_index_build behind the build
        x_range, buckets = index_request[0], int(index_request[1])
        if x_range is not None and C == 1 and F % 16 == 0 and tuple(x_range.shape) == (F, 2) and x_range.dtype == torch.float32 and x_range.is_contiguous():
            index = index_request[2] if len(index_request) > 2 and index_request[2] is not None else (
                torch.empty((F, buckets), dtype=torch.int16, device=dev), torch.empty((F, 2), dtype=torch.float32, device=dev))
            a.index_range, a.index_table, a.index_key, a.index_buckets = (_lib.ptr(x_range), _lib.ptr(index[0]), _lib.ptr(index[1]),
                                                                          buckets)
            keepalive.append(x_range)
    _lib.check(_lib.lib().gnan_pwl_build(a, _lib.stream_of(anchor)), "gnan_pwl_build")
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture nothing may wait for the device: no read-back.  Whoever replays the graph checks
        # BEFORE each replay (an eager build of the same weights) that the tables still fit the captured look-up
        pending = _PendingTables(meta, None, None, anchor, val, slope, F, C, (dev, F, p.L, p.H, C), keepalive + [scratch])
        pending.index = index
        return pending
    # the ONE device->host copy of the build: into a cached pinned buffer, asynchronously; an event marks its arrival
    pinned = _pinned_meta(dev, F + 2, pinned_slot)      # builds in flight at the same time (TablePrefetch) use their own slots
    pinned.copy_(meta, non_blocking=True)
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream(dev))
    pending = _PendingTables(meta, pinned, done, anchor, val, slope, F, C, (dev, F, p.L, p.H, C), keepalive + [scratch])
    pending.index = index
    return pending if lazy else pending.resolve()


_LAST_PLAN = {}     # (device, F, L, H, C) -> (max pieces, features per group, max group pieces) of the last exact tables


class _PendingTables:
    """Tables whose kernels are queued but whose piece counts have not been read back yet."""

    def __init__(self, meta, pinned, done, anchor, val, slope, F, C, key, keepalive):
        self.meta, self.pinned, self.done = meta, pinned, done
        self.anchor, self.val, self.slope = anchor, val, slope
        self.F, self.C, self.key, self.keepalive = F, C, key, keepalive

    def join(self, stream) -> None:
        """Make ``stream`` wait (on the device) for this build if it was queued on another stream, and tell the allocator
        that the tables are used there."""
        owner = getattr(self, "owner_stream", None)
        if owner is None or owner == stream or self.done is None:     # (built inside a captured graph: ordered by its joins)
            return
        stream.wait_event(self.done)
        for t in (self.meta, self.anchor, self.val, self.slope):
            t.record_stream(stream)

    def speculative(self) -> Optional[PwlTables]:
        """Tables sized from the LAST forward's piece counts (same power-of-two search depth, 1/8 more room per feature
        group), usable to queue the look-up before this build's counts are known; None on the first call.  The caller
        must check :meth:`PwlTables` from :meth:`resolve` against them (:func:`covers`) and re-launch if they fall short."""
        last = _LAST_PLAN.get(self.key)
        if last is None:
            return None
        biggest, fg, mg = last
        p2 = 1
        while p2 < biggest:
            p2 <<= 1
        room = min(mg + mg // 8 + 8, LDS_LIMIT // ((1 + 2 * table_stride(self.C)) * 4))     # never ask for more LDS than the kernel accepts
        if room < mg:
            if self.C == 1 or fg != 1:
                return None
            room = mg + mg // 8 + 8      # tables no LDS image holds (C > ~110): the two-phase kernels read them from global memory
        return PwlTables(self.meta[: self.F + 1], self.anchor, self.val, self.slope, max(p2, 64), fg, room)

    def resolve(self) -> Optional[PwlTables]:
        self.done.synchronize()
        host = self.pinned.tolist()
        off_host, overflowed = host[:-1], bool(host[-1])
        biggest = max(b - a_ for a_, b in zip(off_host, off_host[1:]))
        if overflowed or biggest > MAX_PIECES:
            _LAST_PLAN.pop(self.key, None)
            return None
        plan = _plan_groups(off_host, self.C)
        if plan is None:
            _LAST_PLAN.pop(self.key, None)
            return None
        _LAST_PLAN[self.key] = (biggest, plan[0], plan[1])
        n = off_host[-1]
        return PwlTables(self.meta[: self.F + 1], self.anchor[:n], self.val[:n], self.slope[:n], biggest, plan[0], plan[1])


def covers(spec: PwlTables, exact: PwlTables) -> bool:
    """Did a look-up launched with the speculative sizes ``spec`` see every piece of the ``exact`` tables?  It did iff the
    search depth reaches the largest feature and the LDS image holds the largest group of ``spec``'s grouping."""
    if exact.max_pieces > spec.max_pieces:
        return False
    if exact.features_per_group == spec.features_per_group:
        return exact.max_group_pieces <= spec.max_group_pieces
    # different grouping: a group of fg features holds at most fg * max_pieces pieces
    return spec.features_per_group * exact.max_pieces <= spec.max_group_pieces


BUILD_BACKEND = "auto"       # "auto": HIP kernel where it applies, else the (graph-replayed) torch restatement; "torch"


def hip_build_applies(p) -> bool:
    return BUILD_BACKEND == "auto" and p.w_last.is_cuda and p.L in (2, 3) and 1 <= p.H <= 128


@torch.no_grad()
def build_tables_lazy(p, pinned_slot: int = 0, buffers=None, index_request=None):
    """Queue the table build and return a :class:`_PendingTables` (kernel route only; check :func:`hip_build_applies`).
    ``index_request = (x_range [F, 2], buckets[, (table, key)])``: the look-up's direct-index tables come out of the same launches
    (``pending.index``; None where they do not apply)."""
    return _build_tables_hip(p, lazy=True, pinned_slot=pinned_slot, buffers=buffers, index_request=index_request)


@torch.no_grad()
def build_tables(p, use_graph: bool = True) -> Optional[PwlTables]:
    """Tabulate all F shape functions; returns None if some feature needs more than MAX_PIECES pieces."""
    if hip_build_applies(p):
        return _build_tables_hip(p)
    packed, anchor, val, sl, keep = _GraphedBuild.run(p) if use_graph else _build_padded(p)
    C = val.shape[-1]
    host = packed.tolist()                                          # the ONE device->host copy of the build
    off_host, overflowed = host[:-1], bool(host[-1])
    if overflowed or max(b - a for a, b in zip(off_host, off_host[1:])) > MAX_PIECES:
        return None
    plan = _plan_groups(off_host, C)
    if plan is None:
        return None
    sel = keep.reshape(-1).nonzero().squeeze(1)                     # slots that hold a real piece, feature-major
    off = torch.tensor(off_host, dtype=torch.int32).to(anchor.device, non_blocking=True)
    return PwlTables(off, anchor.reshape(-1)[sel].contiguous(), val.reshape(-1, C)[sel].contiguous(),
                     sl.reshape(-1, C)[sel].contiguous(), max(b - a for a, b in zip(off_host, off_host[1:])),
                     plan[0], plan[1])


def evaluate_reference(x: torch.Tensor, t: PwlTables, sum_features: bool) -> torch.Tensor:
    """Plain-torch evaluation of the tables (used by the CPU tests of the table builder; the product path
    evaluates them with ``gnan_fpwl_fwd``)."""
    n, F = x.shape
    C = t.val.shape[1]
    out = x.new_zeros((n, C) if sum_features else (n, F * C))
    off = t.off.tolist()
    for k in range(F):
        a = t.anchor[off[k]:off[k + 1]]
        i = torch.searchsorted(a[1:].contiguous(), x[:, k].contiguous(), right=True)
        y = t.val[off[k]:off[k + 1]][i] + t.slope[off[k]:off[k + 1]][i] * (x[:, k] - a[i]).unsqueeze(1)
        if sum_features:
            out += y
        else:
            out[:, k * C:(k + 1) * C] = y
    return out


# =============================================================================
# backward: per-piece moments -> exact parameter gradients
# =============================================================================
def piece_probe_points(t: PwlTables):
    """Two points strictly inside every piece, ``u1 = a + h`` and ``u2 = a + 2h`` (``h`` = a third of the piece
    width; ``-1`` / ``+1`` on the two unbounded pieces), and ``h`` itself.  All ``[T]`` float64."""
    off = t.off.long()
    T = t.anchor.numel()
    a = t.anchor.double()
    first = torch.zeros(T, dtype=torch.bool, device=a.device)
    last = torch.zeros(T, dtype=torch.bool, device=a.device)
    first[off[:-1]] = True
    last[off[1:] - 1] = True
    nxt = torch.cat([a[1:], a[-1:]])
    w = nxt - a
    h = torch.where(w > 0, w / 3.0, torch.ones_like(w))
    h = torch.where(first, -torch.ones_like(h), h)
    h = torch.where(last, torch.ones_like(h), h)          # a single-piece (affine) feature is both: +1 wins
    # a piece one float32 step wide holds the nodes with x == a and nothing else (_with_point_pieces): both probes sit
    # ON the anchor, where autograd takes relu'(0) = 0; its slope moment is zero (x - a = 0), h only has to be finite
    up = torch.nextafter(t.anchor, torch.full_like(t.anchor, float("inf")))
    nxt32 = torch.cat([t.anchor[1:], t.anchor[-1:]])
    point = (nxt32 > t.anchor) & (nxt32 <= up) & ~first & ~last
    return torch.where(point, a, a + h), torch.where(point, a, a + 2.0 * h), h


def moments_reference(x: torch.Tensor, g: torch.Tensor, t: PwlTables, sum_features: bool) -> torch.Tensor:
    """Plain-torch restatement of ``gnan_fpwl_moments`` (CPU tests): ``M [T, 2, C]``."""
    n, F = x.shape
    C = t.val.shape[1]
    M = torch.zeros(t.anchor.numel(), 2, C, dtype=torch.float64)
    off = t.off.tolist()
    for k in range(F):
        a = t.anchor[off[k]:off[k + 1]]
        i = torch.searchsorted(a[1:].contiguous(), x[:, k].contiguous(), right=True)
        gk = (g if sum_features else g[:, k * C:(k + 1) * C]).double()
        d = (x[:, k] - a[i]).double().unsqueeze(1)
        M[off[k]:off[k + 1], 0].index_add_(0, i, gk)
        M[off[k]:off[k + 1], 1].index_add_(0, i, gk * d)
    return M


def parameter_grads_from_moments(p, t: PwlTables, M: torch.Tensor, evaluate):
    """Exact gradients of ``sum_n <g_n, f(x_n)>`` w.r.t. the stacked parameters, from the per-piece moments.

    On piece ``t`` (anchor ``a``) ``f(x) = val + slope (x - a)`` with ``val``, ``slope`` functions of the
    parameters, so the objective is ``<val, M0> + <slope, M1>``.  With two probe points inside the piece,
    ``val = 2 f(u1) - f(u2)`` and ``slope = (f(u2) - f(u1)) / h``, i.e. the objective equals
    ``<f(u1), 2 M0 - M1/h> + <f(u2), -M0 + M1/h>``: a weighted sum of network outputs at 2T points, which is
    back-propagated through the (tiny) batched MLP ``evaluate(U, p)`` in float64.
    ``p``: StackedMLP of leaf tensors with ``requires_grad``; returns gradients in the order of its non-None tensors.
    """
    u1, u2, h = piece_probe_points(t)
    M = M.double()
    c2 = -M[:, 0] + M[:, 1] / h.unsqueeze(1)                  # [T, C]
    c1 = M[:, 0] - c2                                         # = 2 M0 - M1/h
    off = t.off.long()
    F = off.numel() - 1
    C = M.shape[2]
    # feature and in-feature index of every table row, without a device->host copy (the table may sit in a buffer of full
    # capacity when it was sized for a hipGraph: rows behind off[F] are not pieces — they get zero weights and a clamped slot)
    T = M.shape[0]
    row = torch.arange(T, device=M.device)
    real = row < off[F]
    feat = torch.searchsorted(off[1:].contiguous(), row, right=True).clamp_(max=F - 1)
    rows = 2 * t.max_pieces                                   # host-known: no device->host copy in the backward
    local = (row - off[:-1][feat]).clamp_(0, t.max_pieces - 1)
    zero = torch.zeros((), dtype=torch.float64, device=M.device)
    u1, u2 = torch.where(real, u1, zero), torch.where(real, u2, zero)
    c1, c2 = torch.where(real.unsqueeze(1), c1, zero), torch.where(real.unsqueeze(1), c2, zero)
    U = torch.zeros(rows, F, dtype=torch.float64, device=M.device)
    W = torch.zeros(rows, F, C, dtype=torch.float64, device=M.device)
    U.index_put_((2 * local, feat), u1, accumulate=True)      # (accumulate: the clamped slots of non-pieces add zeros)
    U.index_put_((2 * local + 1, feat), u2, accumulate=True)
    W.index_put_((2 * local, feat), c1, accumulate=True)
    W.index_put_((2 * local + 1, feat), c2, accumulate=True)
    live = [q for q in p[:6] if q is not None]
    with torch.enable_grad():
        y = evaluate(U, p).view(rows, F, C)                   # float64 through the casts inside `evaluate`
        obj = (y * W).sum()
    return torch.autograd.grad(obj, live, allow_unused=True)
