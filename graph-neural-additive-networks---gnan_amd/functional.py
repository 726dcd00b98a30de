"""Shape functions: autograd-aware host wrappers around the HIP kernels that evaluate the per-feature MLPs (``libgnan_hip.so``).

:func:`feature_mlps` — all per-feature shape functions in one launch (replaces the Python loop GNAN.py:57-62): dispatch between
direct evaluation (``csrc/fmlp.hip``) and the exact piecewise-linear tables (``pwl.py`` + ``csrc/fpwl*.hip``), the launch
wrappers of both, their autograd node and its backward (moments + analytic parameter gradients, or ``gnan_fmlp_bwd``), the
pre-rho row table (``rho_row_lut``) and the row reductions of ``csrc/colsum.hip``.

The package's other operators live next door: ``aggregate`` (:func:`~.aggregate.rho_aggregate`, the rho(distance)-weighted
neighbourhood sum — GNAN.py:65-73 / models.py:368-376 and the per-node loop GNAN.py:159-170), ``small_graph`` (one-launch
forward / backward of small graphs), ``losses`` (the epoch loops' loss step).  ``modules`` wires them into the reference's classes.

Everything here needs device tensors; there is no CPU fallback.
"""
from __future__ import annotations

from typing import NamedTuple, Optional, Sequence


import torch

from . import _lib
from ._cache import TensorKeyedCache


# =============================================================================
# stacked per-feature MLP parameters
# =============================================================================
class StackedMLP(NamedTuple):
    """The reference's F separate ``nn.Sequential`` MLPs (GNAN.py:24-34) stacked over the feature axis."""
    w_first: Optional[torch.Tensor]   # [F, H]
    b_first: Optional[torch.Tensor]   # [F, H]
    w_mid: Optional[torch.Tensor]     # [L-2, F, H, H]
    b_mid: Optional[torch.Tensor]     # [L-2, F, H]
    w_last: torch.Tensor              # [F, C, H]  ([F, C] when L == 1)
    b_last: Optional[torch.Tensor]    # [F, C]
    L: int
    H: int
    C: int
    F: int


def stack_mlps(mlps: Sequence[torch.nn.Sequential]) -> StackedMLP:
    """Stack the Linear layers of F structurally identical MLPs (autograd flows back through the stack)."""
    lin = [[m for m in seq if isinstance(m, torch.nn.Linear)] for seq in mlps]
    F, L = len(lin), len(lin[0])
    has_bias = lin[0][0].bias is not None
    C = lin[0][-1].out_features

    def st(idx, attr, squeeze=False):
        ts = [getattr(layers[idx], attr) for layers in lin]
        out = torch.stack(ts, 0)
        return out[..., 0] if squeeze else out

    if L == 1:
        return StackedMLP(None, None, None, None, st(0, "weight", True), st(0, "bias") if has_bias else None,
                          1, 0, C, F)
    H = lin[0][0].out_features
    w_mid = b_mid = None
    if L > 2:
        w_mid = torch.stack([st(l, "weight") for l in range(1, L - 1)], 0)
        b_mid = torch.stack([st(l, "bias") for l in range(1, L - 1)], 0) if has_bias else None
    return StackedMLP(st(0, "weight", True), st(0, "bias") if has_bias else None, w_mid, b_mid,
                      st(L - 1, "weight"), st(L - 1, "bias") if has_bias else None, L, H, C, F)


def _rows(t: torch.Tensor) -> torch.Tensor:
    """``t [n, w]`` as the kernels want it: unit stride inside a row and rows that do not overlap (``stride(0) >= w``).
    An expanded tensor — e.g. the gradient ``feature_mlps(...).sum(0)`` sends back, strides (0, 1) — passes a plain
    ``stride(1) == 1`` test and is then refused by the kernels' argument checks; it is materialised here instead."""
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        return t.contiguous()
    return t


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().float().contiguous()


FMLP_ALGO = _lib.FMLP_AUTO   # tests flip this to pin one of the three evaluation strategies
PWL_MIN_WORK = 1 << 24       # AUTO tabulates the shape functions once n*F look-ups outweigh the table build
PWL_MIN_WORK_GRAD = 1 << 18  # ... much earlier when a backward pass follows: the moment kernel replaces a full recompute
PWL_MIN_NODES = 1 << 14      # inference on small graphs: the matrix-core kernel beats table build + look-up
SUM_VIA_FEATURES_MAX_NODES = 32768   # table look-up with a feature sum and C > 1 on small graphs: per-feature pass + sum


MOMENTS_GENERAL = False   # A/B aid: the general moment kernel where the C = 1 kernel applies
LOCATE_SORTED = False              # A/B aid: the sorted-array search where the tree search applies


INDEX_FLAGS = 0               # A/B aid: _lib.FPWL_INDEX_HALF_LINES / FPWL_INDEX_BS512 for the direct-index look-up


def _fpwl_flags() -> int:
    """``gnan_fpwl_args.flags`` from this module's switches (the library itself reads no environment variables)."""
    return ((_lib.FPWL_MOMENTS_GENERAL if MOMENTS_GENERAL else 0) | (_lib.FPWL_LOCATE_SORTED if LOCATE_SORTED else 0) | INDEX_FLAGS
            | (_lib.FPWL_ROWS_MOMENTS_LANE_PER_CHANNEL if ROWS_MOMENTS_LANE_PER_CHANNEL else 0))


ROWS_MOMENTS_LANE_PER_CHANNEL = False   # A/B: 33..42 channels with a lane per channel instead of a pair of channels per lane
FPWL_ROWS = True   # several output channels: two-phase look-up (csrc/fpwl_rows.hip)
FPWL_ROWS_MIN_NODES = 32768
# fewer channels: locating the pieces separately costs more than it saves (GNAN_FPWL_ROWS_MIN_C: A/B aid).  10M nodes x 64
# features, look-up, two-phase / thread-per-node: C = 2 4.4 / 3.9 ms, C = 4 4.6 / 5.7, C = 8 5.2 / 6.2, C = 12 7.0 / 9.3,
# C = 32 11.8 / 35.7; arxiv-shaped forward+backward: C = 4 2.91 / 2.84, C = 7 2.77 / 2.86
FPWL_ROWS_MIN_CHANNELS = 6
FPWL_ROWS_MIN_CHANNELS_LARGE = min(4, FPWL_ROWS_MIN_CHANNELS)       # from FPWL_ROWS_LARGE nodes (tree-search locate kernel)
FPWL_ROWS_LARGE = 262144


def _fpwl_args(x: torch.Tensor, t, sum_features: bool, out=None) -> "_lib.FpwlArgs":
    n, F = x.shape
    return _lib.FpwlArgs(x=_lib.ptr(x), n=n, x_stride=x.stride(0), F=F, C=t.val.shape[1], off=_lib.ptr(t.off),
                         anchor=_lib.ptr(t.anchor), val=_lib.ptr(t.val), slope=_lib.ptr(t.slope),
                         max_pieces=t.max_pieces, features_per_group=t.features_per_group,
                         max_group_pieces=t.max_group_pieces, sum_features=int(sum_features),
                         out=_lib.ptr(out), out_stride=0 if out is None else out.stride(0), out_dtype=_lib.GNAN_F32,
                         flags=_fpwl_flags())


def _fpwl_rows_applies(n: int, C: int, t, bins: bool = True) -> bool:
    """Several channels on a large batch: the piece of every (node, feature) is located once (``gnan_fpwl_locate``) and the
    channel work runs with lane = channel — arxiv-shaped C = 40: look-up 2.3 -> 0.42 (+ 0.11 locate) ms, moments 4.5 -> 1.17 ms."""
    # more than 64 channels: in chunks of 64 (forward) / of as many channels as have their 64-bit bins in LDS (backward);
    # tables too large for the LDS image of the thread-per-node kernels (C > ~110) have no other kernel: any batch size
    from .pwl import oversize
    if not FPWL_ROWS or not 1 < C <= 4096:
        return False
    min_c = FPWL_ROWS_MIN_CHANNELS_LARGE if n >= FPWL_ROWS_LARGE else FPWL_ROWS_MIN_CHANNELS
    return oversize(t) or (min_c <= C and n >= FPWL_ROWS_MIN_NODES)


def _fpwl_locate(x: torch.Tensor, t, a):
    n, F = x.shape
    piece = torch.empty((n, F), dtype=torch.int32, device=x.device)
    dx = torch.empty((n, F), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().gnan_fpwl_locate(a, _lib.ptr(piece), _lib.ptr(dx), _lib.stream_of(x)), "gnan_fpwl_locate")
    return piece, dx


INDEX_LOOKUP = True           # one channel, whole 16-feature groups: find the piece by arithmetic (csrc/fpwl_index.hip), not by a search
INDEX_BUCKETS = 512           # cells per feature over the range its values take (82 KB of LDS per 32-feature group)
INDEX_MIN_NODES = 1 << 16     # below, the look-up is latency-bound either way and the range pass would not pay
_RANGE_CACHE = TensorKeyedCache(8)    # feature matrix (object identity + version) -> [F, 2] column minima / maxima
_RANGE_CHURN = {}                     # (n, F) -> consecutive misses of that cache


def _feature_range(x: torch.Tensor) -> Optional[torch.Tensor]:
    """``[F, 2]`` column (min, max) of the caller's feature matrix — the range hint of the direct-index look-up
    (``gnan_feature_range``: one pass over x, kept per tensor object and version like the other derived inputs).  The hint
    only steers speed: the look-up is exact for values outside it.  None when the caller hands a NEW feature matrix every
    call (three misses in a row for a shape): the pass would then cost more than the index saves."""
    if not INDEX_LOOKUP or x.dim() != 2 or x.shape[0] < INDEX_MIN_NODES or x.shape[1] % 16:
        return None
    hit = _RANGE_CACHE.get((x,))
    shape = tuple(x.shape)
    if hit is not None:
        _RANGE_CHURN[shape] = 0
        return _pin_for_capture(hit)
    if _RANGE_CHURN.get(shape, 0) >= 3:
        return None
    _RANGE_CHURN[shape] = _RANGE_CHURN.get(shape, 0) + 1
    xr = _rows(x.detach().float())
    n, F = xr.shape
    rng = torch.empty((F, 2), dtype=torch.float32, device=x.device)
    ws = torch.empty(2 * F, dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().gnan_feature_range(_lib.ptr(xr), n, xr.stride(0), F, _lib.ptr(rng), _lib.ptr(ws), ws.numel() * 4,
                                             _lib.stream_of(x)), "gnan_feature_range")
    return _pin_for_capture(_RANGE_CACHE.put((x,), None, rng))


def _fpwl_index(a: "_lib.FpwlArgs", x: torch.Tensor, t, x_range: torch.Tensor) -> Optional[list]:
    """Build the direct-index tables of ``t`` over ``x_range`` (``gnan_fpwl_index_build``: one small workgroup per feature,
    queued behind the table build) and attach them to the look-up's arguments; returns the tensors to keep alive."""
    n, F = x.shape
    if (x_range is None or a.C != 1 or t.features_per_group != 16 or F % 16 or x.stride(0) % 4 or x.data_ptr() % 16
            or t.max_pieces > 4096 or tuple(x_range.shape) != (F, 2)):
        return None
    table = torch.empty((F, INDEX_BUCKETS), dtype=torch.int16, device=x.device)
    key = torch.empty((F, 2), dtype=torch.float32, device=x.device)
    ia = _lib.FpwlIndexArgs(off=_lib.ptr(t.off), anchor=_lib.ptr(t.anchor), F=F, buckets=INDEX_BUCKETS, range=_lib.ptr(x_range), table=_lib.ptr(table), key=_lib.ptr(key),
                            stats=None)
    _lib.check(_lib.lib().gnan_fpwl_index_build(ia, _lib.stream_of(x)), "gnan_fpwl_index_build")
    a.index_table, a.index_key = _lib.ptr(table), _lib.ptr(key)
    a.index_buckets = INDEX_BUCKETS
    return [table, key]


KEEP_PIECES = True   # C == 1 training: the forward's pieces (a byte each) serve the backward
LOCATED_KEEP_MAX_BYTES = 2 << 30   # (piece, dx) of a forward are kept for its backward pass while they stay below 2 GiB


def _fpwl_launch(x: torch.Tensor, t, sum_features: bool, want_total: bool = False, out_dtype=torch.float32,
                 total_rows: Optional[int] = None, located: Optional[list] = None, x_range: Optional[torch.Tensor] = None,
                 index=None):
    """Evaluate pre-built piecewise-linear tables (``pwl.build_tables``) with ``gnan_fpwl_fwd``.
    With ``want_total`` returns ``(out, total)`` where ``total[w] = sum_n out[n, w]`` comes out of the same pass
    when the kernel's fast path applies (one output channel, whole feature groups), else ``total`` is None."""
    x = x.detach().float()
    x = _rows(x)
    n, F = x.shape
    C = t.val.shape[1]
    if (sum_features and C > 1 and F >= 16 and n < SUM_VIA_FEATURES_MAX_NODES and n * F * C * 4 <= (1 << 30)
            and out_dtype == torch.float32):
        # few nodes, many features, several channels (Cora: 2708 x 1434 x 7): the summing kernel walks all feature groups
        # inside one workgroup per 64-128 nodes — a few dozen workgroups.  (node block, feature group) workgroups fill
        # the GPU instead; the feature sum over the [n, F, C] result is a 100-MB reduction
        per = _fpwl_launch(x, t, False)
        out = per.view(n, F, C).sum(dim=1)
        return (out, None) if want_total else out
    global _ROOM_RESULT
    width = C if sum_features else F * C
    if (_OUT_BUFFER is not None and tuple(_OUT_BUFFER.shape) == (n, width) and _OUT_BUFFER.dtype == out_dtype
            and _OUT_BUFFER.stride(1) == 1 and _OUT_BUFFER.device == x.device):
        out = _OUT_BUFFER                          # the caller's rows (feature_mlps(out=...): one operand filled by several look-ups)
    elif sum_features and _ROOM_REQUEST and out_dtype == torch.float32:
        # room behind the rows for the compact copy of the most listed nodes' rows (append_hot_rows): the aggregation then
        # gathers the copy into place instead of copying the whole operand into a larger buffer first (40 MB at 10M nodes)
        room = torch.empty((n + _ROOM_REQUEST, C), dtype=out_dtype, device=x.device)
        out = room[:n]
        _ROOM_RESULT = (out.data_ptr(), room)
    else:
        out = torch.empty((n, C if sum_features else F * C), dtype=out_dtype, device=x.device)
    if out_dtype == torch.float32 and _fpwl_rows_applies(n, C, t, bins=False):
        a = _fpwl_args(x, t, sum_features, out)
        piece, dx = _fpwl_locate(x, t, a)
        _lib.check(_lib.lib().gnan_fpwl_rows_fwd(a, _lib.ptr(piece), _lib.ptr(dx), _lib.stream_of(x)), "gnan_fpwl_rows_fwd")
        if located is not None and 2 * piece.numel() * 4 <= LOCATED_KEEP_MAX_BYTES:
            located[:] = [piece, dx]                    # the backward pass bins the gradient by the same pieces
        return (out, None) if want_total else out
    a = _lib.FpwlArgs(x=_lib.ptr(x), n=n, x_stride=x.stride(0), F=F, C=C, off=_lib.ptr(t.off),
                      anchor=_lib.ptr(t.anchor), val=_lib.ptr(t.val), slope=_lib.ptr(t.slope),
                      max_pieces=t.max_pieces, features_per_group=t.features_per_group,
                      max_group_pieces=t.max_group_pieces, sum_features=int(sum_features),
                      out=_lib.ptr(out), out_stride=out.stride(0),
                      out_dtype=_lib.GNAN_BF16 if out_dtype == torch.bfloat16 else _lib.GNAN_F32, flags=_fpwl_flags())
    if (index is not None and a.C == 1 and t.features_per_group == 16 and F % 16 == 0 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and t.max_pieces <= 4096):
        # direct-index tables built ahead of time for THESE tables (TablePrefetch(x_range=...): on the side stream)
        a.index_table, a.index_key, a.index_buckets = _lib.ptr(index[0]), _lib.ptr(index[1]), int(index[0].shape[1])
    else:
        index_keep = _fpwl_index(a, x, t, x_range)      # noqa: F841  (alive until the look-up is queued)
    sum_ws = None
    if sum_features and a.index_table:
        # a medium batch with several feature groups: a workgroup per (node block, group), partial sums added in group order
        need = _lib.lib().gnan_fpwl_sum_workspace_bytes(a)
        if need:
            sum_ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)      # (alive until the look-up is queued)
            a.sum_workspace, a.sum_workspace_bytes = _lib.ptr(sum_ws), need
    total = None
    if sum_ws is not None and want_total and C == 1 and n > 0:
        # ... and the pass that adds the groups' partial sums hands back the column sum of the result (the rest bucket's operand):
        # a gnan_colsum over the result otherwise (two more launches of a replayed step)
        total = torch.empty(1, dtype=torch.float32, device=x.device)
        tot_ws = torch.empty((n + 255) // 256, dtype=torch.float64, device=x.device)     # (alive until the look-up is queued)
        a.sum_total, a.sum_total_workspace, a.sum_total_workspace_bytes = _lib.ptr(total), _lib.ptr(tot_ws), tot_ws.numel() * 8
        a.sum_total_arrive = _lib.ptr(arrive_counter(x.device, 1))
        a.total_rows = n if total_rows is None else int(total_rows)
    fpg = t.features_per_group
    if (want_total and not sum_features and C == 1 and fpg % 4 == 0 and F % fpg == 0 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and n > 0):
        total = torch.empty(F, dtype=torch.float32, device=x.device)
        need = _lib.lib().gnan_fpwl_total_workspace_bytes(a)
        ws = torch.empty(need // 8, dtype=torch.float64, device=x.device)
        a.total, a.total_workspace, a.total_workspace_bytes = _lib.ptr(total), _lib.ptr(ws), need
        a.total_rows = n if total_rows is None else int(total_rows)
    if (located is not None and KEEP_PIECES and C == 1 and sum_features and fpg % 4 == 0 and t.max_pieces <= 256 and n > 0
            and n * F <= LOCATED_KEEP_MAX_BYTES):
        # training, one channel: the fast feature-sum kernel also stores the piece of every look-up (one byte each) and the
        # moment kernel of the backward pass skips its search (C4 training step: look-up 1.11 -> 1.22 ms, moments 1.25 -> 0.95 ms)
        piece8 = torch.empty(((F + fpg - 1) // fpg, n, fpg), dtype=torch.uint8, device=x.device)    # group-major
        a.piece_out = _lib.ptr(piece8)
        rc = _lib.lib().gnan_fpwl_fwd(a, _lib.stream_of(x))
        if rc == 0:
            located[:] = [piece8]
            return (out, total) if want_total else out
        if rc != _lib.ERR_UNSUPPORTED:                  # only "not this kernel's shape" is a reason to ask again: nothing was launched
            _lib.check(rc, "gnan_fpwl_fwd")
        a.piece_out = None                              # another kernel serves this shape: plain look-up
    _lib.check(_lib.lib().gnan_fpwl_fwd(a, _lib.stream_of(x)), "gnan_fpwl_fwd")
    return (out, total) if want_total else out


_OUT_BUFFER = None            # feature_mlps(out=...): where the table look-up stores its result instead of a fresh tensor
_ROOM_REQUEST = 0             # rows of room feature_mlps(room_rows=...) asks the table look-up to leave behind its [n, C] result
_ROOM_RESULT = None           # (data_ptr of the result, the larger buffer it heads) of the last look-up that did
CAPTURED_BUILDS = []          # (sizes a look-up was captured with, the stacked weights it tabulates) of the capture in progress
CAPTURE_PINS = None           # list while graphed.GraphedCallable captures: cache-owned tensors the captured step reads
CAPTURE_GUARD = None          # float32 [1] while a guarded step is captured: set to 1 by a look-up whose tables outgrew its sizes
CAPTURE_SCRATCH = None        # int32 [64K], ZEROED EAGERLY by graphed.GraphedCallable before its capture and owned by that step:
                              # workspaces whose invariant is "zero between launches" (the small-graph kernel's arrival counter)


ARRIVE_COUNTERS = True        # lend the passes that end in a sum over their workgroups an arrival counter (include/gnan_hip.h):
                              # the last workgroup takes the sum, one launch less per pass
ARRIVE_SLOTS = 64             # counters at the END of the zeroed scratch: slot 0 moment scales, 1 group-sum total, 2 packed rows' q
_ARRIVE_POOL = {}             # (device index, stream handle) -> zeroed int32 [ARRIVE_SLOTS]: eager launches of one stream take turns


def arrive_counter(dev, slot: int):
    """Four bytes that are zero between launches (``gnan::last_block`` leaves them zero), or None where nobody guarantees that:
    a capture that is not the framework's own (``graphed.GraphedCallable`` zeroes its scratch eagerly and owns it)."""
    if not ARRIVE_COUNTERS:
        return None
    if torch.cuda.is_current_stream_capturing():
        ws = CAPTURE_SCRATCH
        if ws is None or ws.device != dev:
            return None
        return ws[ws.numel() - ARRIVE_SLOTS + slot:]
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _ARRIVE_POOL.get(key)
    if buf is None:
        buf = _ARRIVE_POOL[key] = torch.zeros(ARRIVE_SLOTS, dtype=torch.int32, device=dev)
    return buf[slot:]


def _pin_for_capture(obj):
    if CAPTURE_PINS is not None:
        CAPTURE_PINS.append(obj)
    return obj

SPECULATIVE_LOOKUP = True   # queue the look-up before the piece counts are read back
MOMENTS_FIXED_POINT = True    # accumulate the per-piece moments in 64-bit fixed point (integer LDS atomics, reproducible)
_ABS_MAX_CACHE = TensorKeyedCache(16)   # feature matrix (object identity + version) -> device scalar max |x|


def _abs_max_cached(x: torch.Tensor) -> torch.Tensor:
    hit = _ABS_MAX_CACHE.get((x,))
    if hit is None:
        hit = _ABS_MAX_CACHE.put((x,), None, x.abs().max().double())
    return _pin_for_capture(hit)


def _fpwl_moments(x: torch.Tensor, t, grad: torch.Tensor, sum_features: bool,
                  x_abs_max: Optional[torch.Tensor] = None, raw: bool = False, located=None):
    """Per-piece moments of the upstream gradient (``gnan_fpwl_moments[_fixed]``) -> ``[T, 2, C]`` float32; with ``raw``
    the fixed-point route returns ``(moments int64 [T, 2, C], scales float64 [2])`` undivided (``gnan_fpwl_param_grads``
    takes them as they are).

    Fixed-point route (default where the 64-bit bins fit LDS): every term is added as ``round(v * 2^e)`` with ``e``
    chosen on the device from ``max |grad|`` and ``max |x - anchor|`` such that n terms cannot overflow 62 bits —
    per-term resolution 2^-(61 - log2 n) of the largest term, i.e. far below fp32 — and the sums do not depend on
    the order of the atomics."""
    x = x.detach().float()
    x = _rows(x)
    grad = grad.detach().float()
    grad = _rows(grad)
    n, F = x.shape
    C = t.val.shape[1]
    T = t.anchor.numel()
    a = _lib.FpwlArgs(x=_lib.ptr(x), n=n, x_stride=x.stride(0), F=F, C=C, off=_lib.ptr(t.off),
                      anchor=_lib.ptr(t.anchor), val=_lib.ptr(t.val), slope=_lib.ptr(t.slope),
                      max_pieces=t.max_pieces, features_per_group=t.features_per_group,
                      max_group_pieces=t.max_group_pieces, sum_features=int(sum_features), out=None, out_stride=0,
                      flags=_fpwl_flags())
    mgp = t.max_group_pieces
    rows = _fpwl_rows_applies(n, C, t)                  # several channels, large batch: per-feature bins, lane = channel
    if MOMENTS_FIXED_POINT and n > 0 and (rows or (mgp + 1) // 2 * 8 + mgp * (2 * C + 1 if C > 1 else 2) * 8 <= 150 * 1024):
        bits = min(50, 61 - max(1, (max(n, 2) - 1).bit_length()))   # a bin receives at most n terms; a term stays below 2^51
        if x_abs_max is None:
            x_abs_max = x.abs().max().double()
        # scales[0] = 2^floor(bits - log2 max|grad|), scales[1] = 2^floor(bits - log2(max|grad| max|x - anchor|)):
        # one pass over the gradient on the device (gnan_fpwl_moment_scales), no host round trip
        ws_words = _lib.MOMENT_SCALES_WORKSPACE_BYTES // 8
        scales = torch.empty(2 + ws_words, dtype=torch.float64, device=x.device)        # [2] scales | the pass's workspace
        Mi = torch.empty((T, 2, C), dtype=torch.int64, device=x.device)                  # (cleared by the same pass)
        # (tables captured into a hipGraph sit in a buffer of full capacity: only the first off[F] anchors are real)
        sa = _lib.MomentScalesArgs(grad=_lib.ptr(grad), n=n, width=grad.shape[1], bits=bits, grad_stride=grad.stride(0),
                                   anchor=_lib.ptr(t.anchor), T=T, n_anchors=_lib.ptr(t.off[F:]), x_abs_max=_lib.ptr(x_abs_max),
                                   workspace=_lib.ptr(scales[2:]), workspace_bytes=_lib.MOMENT_SCALES_WORKSPACE_BYTES,
                                   scales=_lib.ptr(scales), zero=_lib.ptr(Mi), zero_bytes=Mi.numel() * 8,
                                   arrive_counter=_lib.ptr(arrive_counter(x.device, 0)))
        _lib.check(_lib.lib().gnan_fpwl_moment_scales(sa, _lib.stream_of(x)), "gnan_fpwl_moment_scales")
        scales = scales[:2]
        if located and len(located) == 1 and not rows:          # one channel: the forward's pieces, one byte per look-up
            a.piece_in = _lib.ptr(located[0])
        if rows:
            piece, dx = located if (located and len(located) == 2) else _fpwl_locate(x, t, a)    # kept by the forward pass, or located again
            _lib.check(_lib.lib().gnan_fpwl_rows_moments_fixed(a, _lib.ptr(piece), _lib.ptr(dx), _lib.ptr(grad), grad.stride(0),
                                                               _lib.ptr(scales), _lib.ptr(Mi), _lib.stream_of(x)),
                       "gnan_fpwl_rows_moments_fixed")
        else:
            _lib.check(_lib.lib().gnan_fpwl_moments_fixed(a, _lib.ptr(grad), grad.stride(0), _lib.ptr(scales), _lib.ptr(Mi),
                                                          _lib.stream_of(x)), "gnan_fpwl_moments_fixed")
        if raw:
            return Mi, scales
        return (Mi.double() / scales.view(1, 2, 1)).float()
    M = torch.zeros((T, 2, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().gnan_fpwl_moments(a, _lib.ptr(grad), grad.stride(0), _lib.ptr(M), _lib.stream_of(x)),
               "gnan_fpwl_moments")
    return M


def _fmlp_forward(x: torch.Tensor, p: "StackedMLP", sum_features: bool, want_total: bool = False,
                  needs_grad: bool = False, out_dtype=torch.float32, total_rows: Optional[int] = None, prebuilt=None,
                  located: Optional[list] = None):
    """Strategy choice: exact table look-up for large batches, matrix-core / lane kernel otherwise.
    Returns ``(out, tables or None, total or None)``."""
    algo = FMLP_ALGO
    if out_dtype != torch.float32:                      # bf16 operand rows exist only on the table path
        algo = _lib.FMLP_PWL
    threshold = PWL_MIN_WORK_GRAD if needs_grad else PWL_MIN_WORK
    if algo == _lib.FMLP_PWL or (algo == _lib.FMLP_AUTO and x.shape[0] * p.F >= threshold
                                 and (needs_grad or x.shape[0] >= PWL_MIN_NODES)):
        from .pwl import build_tables, build_tables_lazy, covers, hip_build_applies
        stacked = StackedMLP(*[_c(t) for t in p[:6]], *p[6:])

        x_range = _feature_range(x) if p.C == 1 else None
        built = [getattr(prebuilt, "index", None) if prebuilt is not None else None]      # direct-index tables made with the tables
        request = None if x_range is None else (x_range, INDEX_BUCKETS)

        def look_up(t):
            return _fpwl_launch(x, t, sum_features, want_total=True, out_dtype=out_dtype, total_rows=total_rows,
                                located=located, x_range=x_range, index=built[0]) \
                if want_total else (_fpwl_launch(x, t, sum_features, out_dtype=out_dtype, located=located,
                                                 x_range=x_range, index=built[0]), None)

        if prebuilt is not None and torch.cuda.is_current_stream_capturing():
            # a captured inference step (distributed.SharePipeline): the tables sit in caller-owned buffers the PREVIOUS
            # replay filled; sized like every captured look-up — from the last eager forward's counts — and guarded on the
            # device (the replayer reads the guard, a tripped loop falls back to eager forwards)
            guess = prebuilt.speculative()
            if guess is None:
                raise _lib.GnanHipError("graph capture needs one eager forward of this model first (table sizes unknown)")
            res = look_up(guess)
            CAPTURED_BUILDS.append((guess, stacked))
            if CAPTURE_GUARD is not None and not getattr(prebuilt, "fit_checked", False):
                _lib.check(_lib.lib().gnan_pwl_check_fit(_lib.ptr(prebuilt.meta), stacked.F, int(guess.features_per_group),
                                                         int(guess.max_pieces), int(guess.max_group_pieces),
                                                         _lib.ptr(CAPTURE_GUARD), _lib.stream_of(x)), "gnan_pwl_check_fit")
            return res[0], guess, res[1]
        if prebuilt is not None:
            # tables of THESE weights queued earlier, possibly on another stream (TablePrefetch): wait for that stream's
            # build on the device, then look up as the speculative path does
            prebuilt.join(torch.cuda.current_stream(x.device))
            guess = prebuilt.speculative()
            res = None
            if guess is not None:
                try:
                    res = look_up(guess)
                except _lib.GnanHipError:
                    res = None
            tables = prebuilt.resolve()
            if tables is not None:
                if res is None or not covers(guess, tables):
                    res = look_up(tables)
                return res[0], tables, res[1]
            if algo == _lib.FMLP_PWL:
                raise _lib.GnanHipError("shape functions need more pieces than the look-up kernel supports")
            return _fmlp_launch(x, p, sum_features, _lib.FMLP_AUTO), None, None
        if hip_build_applies(stacked) and torch.cuda.is_current_stream_capturing():
            # hipGraph capture (gnan_amd/graphed.py): no device->host copy may happen here, so the look-up is sized like
            # the speculative one — from the piece counts of the LAST eager forward, with room — and whoever replays the
            # graph checks before every replay that the tables of the current weights still fit (CAPTURED_BUILDS)
            pending = build_tables_lazy(stacked, index_request=request)
            built[0] = pending.index
            guess = pending.speculative()
            if guess is None:
                raise _lib.GnanHipError("graph capture needs one eager forward of this model first (table sizes unknown)")
            res = look_up(guess)
            CAPTURED_BUILDS.append((guess, stacked))
            if CAPTURE_GUARD is not None:
                # the step's guard: set on the device if the tables of the weights a REPLAY finds outgrew these sizes — the
                # captured update is skipped then and the replayer re-runs the step eagerly (graphed.GraphedStep.replay)
                _lib.check(_lib.lib().gnan_pwl_check_fit(_lib.ptr(pending.meta), stacked.F, int(guess.features_per_group),
                                                         int(guess.max_pieces), int(guess.max_group_pieces),
                                                         _lib.ptr(CAPTURE_GUARD), _lib.stream_of(x)), "gnan_pwl_check_fit")
            return res[0], guess, res[1]
        if SPECULATIVE_LOOKUP and hip_build_applies(stacked) and not torch.cuda.is_current_stream_capturing():
            # Sizing the look-up needs the tables' piece counts, i.e. a device->host copy between the table build and the
            # look-up during which the GPU idles (60-70 us: 1 % of the C4 forward, 6 % of a 1/8 share).  The look-up is
            # queued right behind the build with the sizes of the LAST forward plus some room; the counts are read
            # afterwards (they arrived long before) and the look-up is queued again in the rare case they outgrew the guess.
            pending = build_tables_lazy(stacked, index_request=request)
            built[0] = pending.index
            guess = pending.speculative()
            res = None
            if guess is not None:
                try:
                    res = look_up(guess)
                except _lib.GnanHipError:          # a guess the kernel refuses is just a wrong guess
                    res = None
            tables = pending.resolve()
            if tables is not None:
                if res is None or not covers(guess, tables):
                    res = look_up(tables)
                return res[0], tables, res[1]
        else:
            tables = build_tables(stacked)
            if tables is not None:
                out, total = look_up(tables)
                return out, tables, total
        if algo == _lib.FMLP_PWL:
            raise _lib.GnanHipError("shape functions need more pieces than the look-up kernel supports")
    return _fmlp_launch(x, p, sum_features, _lib.FMLP_AUTO if algo == _lib.FMLP_PWL else algo), None, None


def _fmlp_launch(x: torch.Tensor, p: StackedMLP, sum_features: bool, algo: int = 0, dropout=None) -> torch.Tensor:
    x = x.detach().float()
    x = _rows(x)
    n = x.shape[0]
    width = p.C if sum_features else p.F * p.C
    out = torch.empty((n, width), dtype=torch.float32, device=x.device)
    keep = [_c(t) for t in (p.w_first, p.b_first, p.w_mid, p.b_mid, p.w_last, p.b_last)]
    a = _lib.FmlpArgs(x=_lib.ptr(x), n=n, x_stride=x.stride(0), F=p.F, L=p.L, H=p.H, C=p.C,
                      w_first=_lib.ptr(keep[0]), b_first=_lib.ptr(keep[1]), w_mid=_lib.ptr(keep[2]),
                      b_mid=_lib.ptr(keep[3]), w_last=_lib.ptr(keep[4]), b_last=_lib.ptr(keep[5]),
                      sum_features=int(sum_features), out=_lib.ptr(out), out_stride=out.stride(0), algo=algo)
    if dropout is not None:
        a.dropout_p, a.dropout_seed = float(dropout[0]), int(dropout[1])
    need = _lib.lib().gnan_fmlp_fwd_workspace_bytes(a)
    if need:
        ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)   # packed weights (caching allocator)
        a.workspace, a.workspace_bytes = _lib.ptr(ws), need
    _lib.check(_lib.lib().gnan_fmlp_fwd(a, _lib.stream_of(x)), "gnan_fmlp_fwd")
    return out


class TablePrefetch:
    """Software pipeline for inference loops: the tables of the NEXT forward are built on a side stream while the current
    forward's look-up and aggregation run (the build occupies 64-129 workgroups for 0.05 ms and depends on the weights
    only).  ``launch()`` queues a build of the current weights and returns a handle for ``feature_mlps(tables=...)``; the
    consumer's stream waits for the build on the device, never on the host.  Every forward still gets a build of its own —
    nothing is cached across forwards."""

    def __init__(self, stacked: StackedMLP):
        from .pwl import hip_build_applies
        self.stacked = StackedMLP(*[_c(t) for t in stacked[:6]], *stacked[6:])
        self.applies = hip_build_applies(self.stacked)
        self.side = torch.cuda.Stream(device=self.stacked.w_last.device) if self.applies else None

    def launch(self, buffers=None, x_range=None, index_buffers=None, guard=None):
        """``buffers``: caller-owned outputs (``pwl.table_buffers``) instead of fresh ones — what a captured loop needs.
        ``x_range`` (+ ``index_buffers = (table [F, B] int16, key [F, 2] float32)``): the direct-index tables of the look-up
        are built right behind the tables, on the side stream too.  ``guard``: the tables are checked against the
        speculative look-up sizes there as well (``gnan_pwl_check_fit``; a captured loop reads the flag afterwards).
        Under a hipGraph capture all of it becomes a forked branch of the graph: the caller joins it
        (``current_stream.wait_stream(prefetch.side)``) before the capture ends."""
        if not self.applies:
            return None
        from .pwl import build_tables_lazy
        main = torch.cuda.current_stream(self.stacked.w_last.device)
        self.side.wait_stream(main)               # the weights (and the allocator's reuse of freed blocks) are ordered
        self.slot = 1 + (getattr(self, "slot", 1) % 2)          # two builds can be in flight: alternate read-back buffers
        with torch.cuda.stream(self.side):
            request = None
            if x_range is not None and index_buffers is not None:
                request = (x_range, int(index_buffers[0].shape[1]), index_buffers)
            pending = build_tables_lazy(self.stacked, pinned_slot=self.slot, buffers=buffers, index_request=request)
            F = self.stacked.F
            if request is not None and pending.index is None and self.stacked.C == 1 and F % 16 == 0:
                table, key = index_buffers                   # (pwl.INDEX_IN_BUILD off: the builder's own launch)
                ia = _lib.FpwlIndexArgs(off=_lib.ptr(pending.meta), anchor=_lib.ptr(pending.anchor), F=F,
                                        buckets=int(table.shape[1]), range=_lib.ptr(x_range), table=_lib.ptr(table),
                                        key=_lib.ptr(key), stats=None)
                _lib.check(_lib.lib().gnan_fpwl_index_build(ia, _lib.stream_of(table)), "gnan_fpwl_index_build")
                pending.index = (table, key)
            if guard is not None:
                spec = pending.speculative()
                if spec is not None:
                    _lib.check(_lib.lib().gnan_pwl_check_fit(_lib.ptr(pending.meta), F, int(spec.features_per_group),
                                                             int(spec.max_pieces), int(spec.max_group_pieces),
                                                             _lib.ptr(guard), _lib.stream_of(guard)), "gnan_pwl_check_fit")
                    pending.fit_checked = True
        pending.owner_stream = self.side
        return pending


HIP_TABLE_GRADS = True   # table path: parameter gradients by gnan_fpwl_param_grads


def _table_grads_applies(L: int, H: int, C: int) -> bool:
    return HIP_TABLE_GRADS and C <= 4096 and ((L == 3 and H <= 64) or (L == 2 and H <= 128))


def _grad_outputs(keep, dests):
    """Output tensors of a parameter-gradient kernel: the caller's destinations (``FlatMLPStore.grad_dest`` — the kernel then
    writes straight into the flat gradient buffer the Parameters' ``.grad`` views are cut from, no copy) where they fit."""
    outs = []
    for i, q in enumerate(keep):
        if q is None:
            outs.append(None)
            continue
        claim = None if dests is None else dests[i]
        d = None if claim is None else claim(q.shape, q.device)      # claimed only if it fits: a claim must be used
        outs.append(d if d is not None else torch.empty_like(q))
    return outs


def _fpwl_param_grads_launch(params, t, moments, L, H, C, F, dests=None):
    """``gnan_fpwl_param_grads``: gradients of the six stacked parameter tensors (None where a bias is absent) from the
    per-piece moments — ``moments`` is ``[T, 2, C]`` float32 or the ``(int64 moments, scales)`` pair of the fixed-point route."""
    fixed = isinstance(moments, tuple)
    if C > 64:
        # the kernel covers 64 output channels.  Its result is linear in (moments, last-layer rows) channel by channel — the
        # activation masks depend on the hidden layers only — so chunks of channels are run one after the other: the last
        # layer's gradients are theirs, the hidden layers' gradients add up
        total = None
        for c0 in range(0, C, 64):
            c1 = min(C, c0 + 64)
            sub = list(params)
            sub[4] = params[4][:, c0:c1].contiguous()
            sub[5] = None if params[5] is None else params[5][:, c0:c1].contiguous()
            m = (moments[0][:, :, c0:c1].contiguous(), moments[1]) if fixed else moments[:, :, c0:c1].contiguous()
            got = _fpwl_param_grads_launch(sub, t, m, L, H, c1 - c0, F)
            if total is None:
                total = [None if q is None else q.clone() for q in got]
                total[4] = torch.empty_like(params[4], dtype=torch.float32)
                total[5] = None if params[5] is None else torch.empty_like(params[5], dtype=torch.float32)
            else:
                for i in range(4):
                    if total[i] is not None:
                        total[i] += got[i]
            total[4][:, c0:c1] = got[4]
            if total[5] is not None:
                total[5][:, c0:c1] = got[5]
        return total
    keep = [None if q is None else q.detach().float().contiguous() for q in params]
    outs = _grad_outputs(keep, dests)
    M = None if fixed else moments.detach().float().contiguous()
    a = _lib.FpwlGradArgs(
        off=_lib.ptr(t.off), anchor=_lib.ptr(t.anchor), moments=_lib.ptr(M),
        moments_fixed=_lib.ptr(moments[0]) if fixed else None, scales=_lib.ptr(moments[1]) if fixed else None,
        w_first=_lib.ptr(keep[0]), b_first=_lib.ptr(keep[1]),
        w_mid=None if keep[2] is None else _lib.ptr(keep[2][0]), b_mid=None if keep[3] is None else _lib.ptr(keep[3][0]),
        w_last=_lib.ptr(keep[4]), b_last=_lib.ptr(keep[5]), F=F, L=L, H=H, C=C, max_pieces=int(t.max_pieces),
        d_w_first=_lib.ptr(outs[0]), d_b_first=_lib.ptr(outs[1]),
        d_w_mid=None if outs[2] is None else _lib.ptr(outs[2][0]), d_b_mid=None if outs[3] is None else _lib.ptr(outs[3][0]),
        d_w_last=_lib.ptr(outs[4]), d_b_last=_lib.ptr(outs[5]))
    _lib.check(_lib.lib().gnan_fpwl_param_grads(a, _lib.stream_of(t.anchor)), "gnan_fpwl_param_grads")
    return outs


HIP_SMALL_BACKWARD = True
# gnan_fmlp_bwd recomputes the activations of every (node, feature) pair (3 H^2 fmas each, fp32 vector units): beyond a few
# million pairs the batched GEMMs of the torch route (matrix cores) catch up; AUTO sends such sizes to the table route anyway
HIP_SMALL_BACKWARD_MAX_WORK = 1 << 23


def _fmlp_backward_launch(x, params, grad_out, sum_features, L, H, C, F, dropout=None, dests=None):
    """``gnan_fmlp_bwd``: gradients of the six stacked parameter tensors (None where a bias is absent), in their order."""
    xd = x.detach().float()
    xd = _rows(xd)
    g = grad_out.detach().float()
    g = _rows(g)
    keep = [None if t is None else t.detach().float().contiguous() for t in params]
    outs = _grad_outputs(keep, dests)
    w_mid = None if keep[2] is None else keep[2][0]              # [1, F, H, H] -> [F, H, H]   (absent for L == 2)
    d_w_mid = None if outs[2] is None else outs[2][0]
    b_mid = None if keep[3] is None else keep[3][0]
    d_b_mid = None if outs[3] is None else outs[3][0]
    a = _lib.FmlpBwdArgs(x=_lib.ptr(xd), n=xd.shape[0], x_stride=xd.stride(0), F=F, L=L, H=H, C=C,
                         w_first=_lib.ptr(keep[0]), b_first=_lib.ptr(keep[1]), w_mid=_lib.ptr(w_mid), b_mid=_lib.ptr(b_mid),
                         w_last=_lib.ptr(keep[4]), b_last=_lib.ptr(keep[5]), sum_features=int(sum_features),
                         grad=_lib.ptr(g), grad_stride=g.stride(0),
                         d_w_first=_lib.ptr(outs[0]), d_b_first=_lib.ptr(outs[1]), d_w_mid=_lib.ptr(d_w_mid),
                         d_b_mid=_lib.ptr(d_b_mid), d_w_last=_lib.ptr(outs[4]), d_b_last=_lib.ptr(outs[5]))
    if dropout is not None:
        a.dropout_p, a.dropout_seed = float(dropout[0]), int(dropout[1])
    need = _lib.lib().gnan_fmlp_bwd_workspace_bytes(a)
    if need:
        ws = torch.empty(need // 4, dtype=torch.float32, device=xd.device)
        a.workspace, a.workspace_bytes = _lib.ptr(ws), need
    _lib.check(_lib.lib().gnan_fmlp_bwd(a, _lib.stream_of(xd)), "gnan_fmlp_bwd")
    return outs


def _fmlp_eager(x: torch.Tensor, p: StackedMLP, sum_features: bool, dropout: float = 0.0) -> torch.Tensor:
    """Batched-GEMM restatement on the device.  Used (a) inside backward passes to obtain parameter gradients
    (recompute + torch autograd on small batches; two probe points per piece on the table path) and (b) as the
    forward while training-mode Dropout is active (``dropout > 0``: ``F.dropout`` after every hidden ReLU,
    GNAN.py:28,32).  The eval-mode forward never comes here."""
    drop = (lambda h: torch.nn.functional.dropout(h, dropout, training=True)) if dropout > 0 else (lambda h: h)
    xt = x.t().unsqueeze(-1)                                            # [F, n, 1]
    if p.L == 1:
        h = xt * p.w_last.unsqueeze(1)                                  # [F, n, C]
        if p.b_last is not None:
            h = h + p.b_last.unsqueeze(1)
    else:
        h = xt * p.w_first.unsqueeze(1)
        if p.b_first is not None:
            h = h + p.b_first.unsqueeze(1)
        h = drop(torch.relu(h))                                         # [F, n, H]
        for l in range(p.L - 2):
            h = torch.bmm(h, p.w_mid[l].transpose(1, 2))
            if p.b_mid is not None:
                h = h + p.b_mid[l].unsqueeze(1)
            h = drop(torch.relu(h))
        h = torch.bmm(h, p.w_last.transpose(1, 2))                      # [F, n, C]
        if p.b_last is not None:
            h = h + p.b_last.unsqueeze(1)
    if sum_features:
        return h.sum(0)                                                 # [n, C]
    return h.permute(1, 0, 2).reshape(x.shape[0], -1)                   # [n, F*C]


_BWD_CHUNK_ELEMS = 1 << 28   # activation floats per recompute chunk (1 GiB)


def dropout_masks(seed: int, p: float, n: int, F: int, n_hidden: int, H: int, device) -> torch.Tensor:
    """The keep-mask the kernels apply for ``(p, seed)`` (``gnan_dropout_mask``): uint8 ``[n, F, n_hidden, H]``."""
    m = torch.empty((n, F, n_hidden, H), dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().gnan_dropout_mask(int(seed), float(p), n, F, n_hidden, H, _lib.ptr(m), _lib.stream_of(m)),
               "gnan_dropout_mask")
    return m


def _fmlp_eager_masked(x: torch.Tensor, p: StackedMLP, sum_features: bool, masks: torch.Tensor, drop_p: float) -> torch.Tensor:
    """:func:`_fmlp_eager` with the kernels' Dropout masks (``dropout_masks``): the backward route of shapes
    ``gnan_fmlp_bwd`` does not cover (deeper / wider than the reference's defaults) under training-mode Dropout."""
    scale = 1.0 / (1.0 - drop_p)
    xt = x.t().unsqueeze(-1)                                            # [F, n, 1]
    h = xt * p.w_first.unsqueeze(1)
    if p.b_first is not None:
        h = h + p.b_first.unsqueeze(1)
    h = torch.relu(h) * (masks[:, :, 0].permute(1, 0, 2).to(h.dtype) * scale)
    for l in range(p.L - 2):
        h = torch.bmm(h, p.w_mid[l].transpose(1, 2))
        if p.b_mid is not None:
            h = h + p.b_mid[l].unsqueeze(1)
        h = torch.relu(h) * (masks[:, :, l + 1].permute(1, 0, 2).to(h.dtype) * scale)
    h = torch.bmm(h, p.w_last.transpose(1, 2))
    if p.b_last is not None:
        h = h + p.b_last.unsqueeze(1)
    return h.sum(0) if sum_features else h.permute(1, 0, 2).reshape(x.shape[0], -1)


class _DropoutMLPs(torch.autograd.Function):
    """The shape functions with training-mode Dropout behind every hidden ReLU (GNAN.py:28,32; run.sh trains with p = 0.6),
    in the kernels: forward ``gnan_fmlp_fwd`` (lane kernel), backward ``gnan_fmlp_bwd`` — both recompute the masks from
    ``(p, seed)`` (csrc/dropout.hpp), no mask or activation tensor is formed.  Shapes the backward kernel does not cover
    (L > 3, H > 64, C > 8) back-propagate through the batched restatement in chunks of nodes, with the kernels' masks."""

    @staticmethod
    def forward(ctx, x, sum_features, drop_p, seed, L, H, C, F, *params):
        p = StackedMLP(*params, L, H, C, F)
        ctx.meta = (sum_features, float(drop_p), int(seed), L, H, C, F)
        ctx.present = [t is not None for t in params]
        ctx.save_for_backward(x, *[t for t in params if t is not None])
        return _fmlp_launch(x, p, sum_features, _lib.FMLP_LANE, dropout=(drop_p, seed))

    @staticmethod
    def backward(ctx, grad_out):
        sum_features, drop_p, seed, L, H, C, F = ctx.meta
        saved = list(ctx.saved_tensors)
        x = saved.pop(0)
        params = [saved.pop(0) if present else None for present in ctx.present]
        if L in (2, 3) and 1 <= H <= 64 and C <= 8 and not ctx.needs_input_grad[0]:
            pg = _fmlp_backward_launch(x, params, grad_out, sum_features, L, H, C, F, dropout=(drop_p, seed))
            return (None,) * 8 + tuple(pg)
        leaves = [None if t is None else t.detach().requires_grad_(True) for t in params]
        p = StackedMLP(*leaves, L, H, C, F)
        live = [t for t in leaves if t is not None]
        grads = [torch.zeros_like(t) for t in live]
        n = x.shape[0]
        chunk = max(1, _BWD_CHUNK_ELEMS // max(1, F * max(H, C) * 2))
        xd = x.detach().float()
        gx = torch.zeros_like(xd) if ctx.needs_input_grad[0] else None
        masks = dropout_masks(seed, drop_p, n, F, L - 1, H, x.device) if n * F * (L - 1) * H <= (1 << 31) else None
        for lo in range(0, n, chunk):
            xs = xd[lo:lo + chunk]
            if gx is not None:
                xs = xs.clone().requires_grad_(True)
            if masks is not None:
                mk = masks[lo:lo + chunk]
            else:                       # (the hash takes the global node index: a chunk's masks are cut from a window's)
                raise _lib.GnanHipError("training-mode Dropout: this shape's backward needs more than 2 GiB of masks")
            with torch.enable_grad():
                out = _fmlp_eager_masked(xs, p, sum_features, mk, drop_p)
            got = torch.autograd.grad(out, live + ([xs] if gx is not None else []), grad_out[lo:lo + chunk])
            for g, d in zip(grads, got):
                g += d
            if gx is not None:
                gx[lo:lo + chunk] = got[-1]
        it = iter(grads)
        pg = [next(it) if present else None for present in ctx.present]
        return (gx, None, None, None, None, None, None, None, *pg)


def feature_mlps_dropout(x: torch.Tensor, p: StackedMLP, sum_features: bool, drop_p: float, seed: Optional[int] = None,
                         return_total: bool = False):
    """:func:`feature_mlps` in training mode with Dropout ``drop_p`` behind every hidden ReLU (GNAN.py:28,32).  ``seed``:
    64 bits that fix the masks; by default drawn from torch's CPU generator (so ``torch.manual_seed`` makes a run
    repeatable, as it does for ``nn.Dropout``).  The masks are a function of (seed, node, feature, layer, unit) —
    :func:`dropout_masks` — not torch's Bernoulli stream."""
    _lib.require_device(x, p.w_last)
    if x.shape[1] != p.F:
        raise ValueError(f"x has {x.shape[1]} feature columns, the model was built for {p.F}")
    if not 0.0 < drop_p < 1.0:
        raise ValueError("drop_p must be in (0, 1)")
    if p.L == 1:                                         # a single Linear has no hidden ReLU: nothing to drop (GNAN.py:25-26)
        return feature_mlps(x, p, sum_features, return_total=return_total)
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
    out = _DropoutMLPs.apply(x, sum_features, float(drop_p), int(seed) & ((1 << 63) - 1), p.L, p.H, p.C, p.F, *p[:6])
    return (out, column_sums(out)) if return_total else out


class _FeatureMLPs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sum_features, want_total, out_dtype, total_rows, L, H, C, F, *params):
        p = StackedMLP(*params, L, H, C, F)
        ctx.meta = (sum_features, L, H, C, F)
        ctx.save_for_backward(x, *[t for t in params if t is not None])
        ctx.present = [t is not None for t in params]
        ctx.grad_dests = _grad_dests_of(params)
        ctx.set_materialize_grads(False)      # (the non-differentiable column sums would get a zero-filled gradient: a launch)
        needs_grad = any(ctx.needs_input_grad[9:]) and not ctx.needs_input_grad[0]
        if needs_grad and out_dtype != torch.float32:
            raise _lib.GnanHipError("bf16 operand storage is an inference format: no backward pass")
        if total_rows is not None and not 0 <= total_rows <= x.shape[0]:
            raise ValueError(f"total_rows={total_rows} outside [0, {x.shape[0]}]")
        fused_total = want_total and total_rows != 0           # 0 rows: the kernel reads that as "all", sum nothing instead
        ctx.located = [] if needs_grad else None
        out, ctx.tables, total = _fmlp_forward(x, p, sum_features, fused_total, needs_grad, out_dtype, total_rows,
                                               located=ctx.located)
        # max |x| scales the fixed-point moments of the backward pass; it is looked up here because this is where the
        # caller's own tensor object is in hand (backward sees a fresh unpacked copy of the saved tensor every time)
        ctx.x_abs_max = _abs_max_cached(x) if (needs_grad and ctx.tables is not None and x.numel()) else None
        if not want_total:
            return out
        if total_rows == 0:
            total = out.new_zeros(out.shape[1], dtype=torch.float32)
        if total is None:
            total = column_sums(out if total_rows is None else out[:total_rows])
        ctx.mark_non_differentiable(total)     # d total / d out is accounted for inside the aggregation's backward
        return out, total

    @staticmethod
    def backward(ctx, grad_out, *unused):
        sum_features, L, H, C, F = ctx.meta
        saved = list(ctx.saved_tensors)
        x = saved.pop(0)
        params = [saved.pop(0) if present else None for present in ctx.present]
        located, ctx.located = ctx.located, None
        if grad_out is None:                  # nothing downstream used the values
            return (None,) * (9 + len(params))
        gx, pg = _shape_function_grads(x, params, ctx.present, ctx.tables, grad_out, sum_features, L, H, C, F, ctx.x_abs_max,
                                       located, ctx.needs_input_grad[0], dests=ctx.grad_dests)
        return (gx, None, None, None, None, None, None, None, None, *pg)


def _grad_dests_of(params):
    """Per stacked parameter tensor: the getter of its direct gradient destination (set by ``FlatMLPStore.stacked``) or None."""
    return [None if t is None else getattr(t, "gnan_grad_dest", None) for t in params]


def _shape_function_grads(x, params, present, tables, grad_out, sum_features, L, H, C, F, x_abs_max=None, located=None,
                          want_x_grad=False, dests=None):
    """``(d x or None, [gradients of the six stacked parameter tensors, None where absent])`` of
    ``sum_n <grad_out[n], f(x[n])>`` — autograd through GNAN.py:57-62 (``sum_features``: through GNAN.py:157 as well)."""
    leaves = [None if t is None else t.detach().requires_grad_(True) for t in params]
    p = StackedMLP(*leaves, L, H, C, F)
    live = [t for t in leaves if t is not None]
    if tables is not None and not want_x_grad:
        # table path: one streaming pass bins the upstream gradient per piece (HIP), then the exact
        # parameter gradients follow from 2 probe points per piece through the tiny batched MLP
        from .pwl import parameter_grads_from_moments
        if _table_grads_applies(L, H, C) and x.is_cuda:
            # ... exactly, in one kernel: one reverse pass for the value and one for the slope of every non-empty piece
            M = _fpwl_moments(x, tables, grad_out, sum_features, x_abs_max, raw=True, located=located)
            return None, _fpwl_param_grads_launch(params, tables, M, L, H, C, F, dests=dests)
        M = _fpwl_moments(x, tables, grad_out, sum_features, x_abs_max, located=located)
        got = parameter_grads_from_moments(
            p, tables, M, lambda U, q: _fmlp_eager(U, StackedMLP(*[None if t is None else t.double()
                                                                  for t in q[:6]], *q[6:]), False))
        it = iter(got)
        pg = [None if not pr else next(it) for pr in present]
        return None, [None if g is None else g.to(torch.float32) for g in pg]
    if (HIP_SMALL_BACKWARD and L in (2, 3) and 1 <= H <= 64 and C <= 8 and not want_x_grad and x.is_cuda
            and 0 < x.shape[0] * F <= HIP_SMALL_BACKWARD_MAX_WORK):
        # small batches (the forward evaluated the MLPs directly): one workgroup per feature recomputes the
        # activations node by node and accumulates every parameter gradient in registers (gnan_fmlp_bwd) — the torch
        # restatement below materialises [F, n, H] activations for the same sums (Cora-shaped: 1 GB per layer)
        return None, _fmlp_backward_launch(x, params, grad_out, sum_features, L, H, C, F, dests=dests)
    grads = [torch.zeros_like(t) for t in live]
    n = x.shape[0]
    chunk = max(1, _BWD_CHUNK_ELEMS // max(1, F * max(H, C)))
    xd = x.detach().float()
    gx = torch.zeros_like(xd) if want_x_grad else None
    for lo in range(0, n, chunk):
        xs = xd[lo:lo + chunk]
        if gx is not None:
            xs = xs.clone().requires_grad_(True)
        with torch.enable_grad():
            out = _fmlp_eager(xs, p, sum_features)
        got = torch.autograd.grad(out, live + ([xs] if gx is not None else []), grad_out[lo:lo + chunk])
        for g, d in zip(grads, got):
            g += d
        if gx is not None:
            gx[lo:lo + chunk] = got[-1]
    it = iter(grads)
    return gx, [next(it) if pr else None for pr in present]


# =============================================================================
# pre-rho normalisation: per-row weight table from the shell counts
# =============================================================================
PRE_RHO_TABLE_MIN = 1 << 16  # (row, hop code) pairs from which rho is tabulated and looked up by gnan_rho_row_lut; below,
                             # the lane / matrix-core shape-function kernels evaluate it (table build + read-back cost more)


def _rho_row_lut_launch(cnt: torch.Tensor, u: torch.Tensor, tables, C: int, want_arg: bool):
    """``(lut [n, D, C], arg [n, D] or None)`` by ``gnan_rho_row_lut``."""
    n, D = cnt.shape
    cnt = cnt if cnt.stride(1) == 1 else cnt.contiguous()
    lut = torch.empty((n, D, C), dtype=torch.float32, device=cnt.device)
    arg = torch.empty((n, D), dtype=torch.float32, device=cnt.device) if want_arg else None
    a = _lib.RhoLutArgs(cnt=_lib.ptr(cnt), cnt_stride=cnt.stride(0), n_rows=n, D=D, C=C, u=_lib.ptr(u),
                        anchor=_lib.ptr(tables.anchor), val=_lib.ptr(tables.val), slope=_lib.ptr(tables.slope),
                        n_pieces=_lib.ptr(tables.off[1:]), max_pieces=int(tables.max_pieces), lut=_lib.ptr(lut),
                        arg=_lib.ptr(arg))
    _lib.check(_lib.lib().gnan_rho_row_lut(a, _lib.stream_of(lut)), "gnan_rho_row_lut")
    return lut, arg


def _rho_param_grads(arg: torch.Tensor, dlut: torch.Tensor, tables, params, present, L: int, H: int, C: int):
    """Gradients of rho's stacked parameters from the gradient of its per-row table: binned by the pieces of the table's
    arguments (one feature), then exact (``gnan_fpwl_param_grads`` / two probe points per piece in float64)."""
    x = arg.reshape(-1, 1)
    g = dlut.reshape(-1, C)
    x_abs_max = x.abs().max().double() if x.numel() else None
    if _table_grads_applies(L, H, C):
        M = _fpwl_moments(x, tables, g, False, x_abs_max, raw=True)
        return tuple(_fpwl_param_grads_launch(params, tables, M, L, H, C, 1))
    from .pwl import parameter_grads_from_moments
    M = _fpwl_moments(x, tables, g, False, x_abs_max)
    leaves = [None if t is None else t.detach().requires_grad_(True) for t in params]
    got = parameter_grads_from_moments(
        StackedMLP(*leaves, L, H, C, 1), tables, M,
        lambda U, q: _fmlp_eager(U, StackedMLP(*[None if t is None else t.double() for t in q[:6]], *q[6:]), False))
    it = iter(got)
    pg = [None if not pr else next(it) for pr in present]
    return tuple(None if t is None else t.to(torch.float32) for t in pg)


class _RhoRowLut(torch.autograd.Function):
    """``lut[i, d, :] = rho(u[d] / max(cnt[i, d], 1))`` from rho's piecewise-linear table (``gnan_rho_row_lut``); backward:
    the gradient of the table binned by the pieces of its arguments (``gnan_fpwl_moments*``, one feature), then the exact
    parameter gradients (``gnan_fpwl_param_grads`` / ``pwl.parameter_grads_from_moments``) — GNAN.py:65-67 under autograd."""

    @staticmethod
    def forward(ctx, cnt, u, tables, L, H, C, *params):
        lut, arg = _rho_row_lut_launch(cnt, u, tables, C, any(ctx.needs_input_grad[6:]))
        ctx.tables, ctx.meta = tables, (L, H, C)
        ctx.present = [t is not None for t in params]
        ctx.save_for_backward(arg, *[t for t in params if t is not None])
        return lut

    @staticmethod
    def backward(ctx, dlut):
        L, H, C = ctx.meta
        saved = list(ctx.saved_tensors)
        arg = saved.pop(0)
        params = [saved.pop(0) if present else None for present in ctx.present]
        return (None,) * 6 + _rho_param_grads(arg, dlut, ctx.tables, params, ctx.present, L, H, C)


def _rho_tables(p: StackedMLP, n_lookups: int):
    """rho's piecewise-linear table when the table route applies to ``n_lookups`` (row, hop code) pairs, else None."""
    if (FMLP_ALGO == _lib.FMLP_PWL or (FMLP_ALGO == _lib.FMLP_AUTO and n_lookups >= PRE_RHO_TABLE_MIN)) \
            and not torch.cuda.is_current_stream_capturing():
        from .pwl import build_tables
        return build_tables(StackedMLP(*[_c(t) for t in p[:6]], *p[6:]))
    return None


def rho_row_lut(cnt: torch.Tensor, u: torch.Tensor, p: StackedMLP) -> torch.Tensor:
    """Per-row weight table of the pre-rho normalisation (GNAN.py:65-67): ``lut[i, d, :] = rho(u[d] / max(cnt[i, d], 1))``,
    ``[n, D, C]``.  ``p``: rho's Linear layers stacked as a ONE-feature StackedMLP; ``u``: the values ``node_distances`` takes
    (``graph.hop_inputs``); ``cnt``: shell counts ``[n, D]`` int32.  Differentiable w.r.t. rho's parameters.  Large tables go
    through rho's exact piecewise-linear tabulation (D look-ups per row, nothing of size N x H is ever formed); small ones
    through the lane / matrix-core kernels of :func:`feature_mlps`."""
    _lib.require_device(cnt, u, p.w_last)
    if p.F != 1:
        raise ValueError("rho is one scalar function: a one-feature StackedMLP is expected")
    n, D = cnt.shape
    tables = _rho_tables(p, n * D)
    if tables is not None:
        return _RhoRowLut.apply(cnt, u.contiguous(), tables, p.L, p.H, p.C, *p[:6])
    arg = (u.unsqueeze(0) / cnt.clamp_min(1).float()).view(-1, 1)              # torch.div, GNAN.py:66
    return feature_mlps(arg, p, False).view(n, D, p.C)


PAD_FEATURES = 16            # the fast look-up / moment kernels and the 16-byte operand gathers want whole 16-feature groups
PAD_MIN_WORK = 1 << 26       # n * F from which a ragged feature count is padded: the arxiv-shaped graph (n * F = 2^24.4) is
                             # bound by the host, the six concatenations (and their backward) cost it 0.1 / 1.5 ms
PAD_MIN_WORK_STORE = 1 << 22  # ... when the parameter store already holds the padded stack (no concatenations): from 4M look-ups
_X_PAD_CACHE = TensorKeyedCache(4)   # feature matrix (object identity + version), padded width -> zero-padded copy


def _padded_x(x: torch.Tensor, Fp: int) -> torch.Tensor:
    if x.requires_grad:
        return torch.nn.functional.pad(x.float(), (0, Fp - x.shape[1]))
    hit = _X_PAD_CACHE.get((x,), Fp)
    if hit is None:
        hit = _X_PAD_CACHE.put((x,), Fp, torch.nn.functional.pad(x.detach().float(), (0, Fp - x.shape[1])))
    return _pin_for_capture(hit)


def _padded_stack(p: StackedMLP, Fp: int) -> StackedMLP:
    """``p`` with ``Fp - F`` all-zero shape functions appended (f = 0; autograd flows to the real ones through the cat)."""
    def pad(t, dim):
        if t is None:
            return None
        shape = list(t.shape)
        shape[dim] = Fp - p.F
        return torch.cat([t, t.new_zeros(shape)], dim=dim)
    return StackedMLP(pad(p.w_first, 0), pad(p.b_first, 0), pad(p.w_mid, 1), pad(p.b_mid, 1), pad(p.w_last, 0),
                      pad(p.b_last, 0), p.L, p.H, p.C, Fp)


def feature_mlps(x: torch.Tensor, p: StackedMLP, sum_features: bool, return_total: bool = False,
                 out_dtype=torch.float32, total_rows: Optional[int] = None, pad_ok: bool = False, tables=None,
                 room_rows: int = 0, out: Optional[torch.Tensor] = None):
    """See :func:`_feature_mlps`.  ``room_rows`` (with ``sum_features``): the ``[n, C]`` result may head a buffer with that
    many more rows — marked by its ``gnan_room`` attribute — which :func:`rho_aggregate` fills with the compact copy of the
    most listed nodes' rows instead of copying the operand (``append_hot_rows``).  ``out`` (inference, table path): rows of a
    caller-owned operand the look-up writes into — the result IS ``out`` when the look-up could use it (same shape and dtype),
    a fresh tensor otherwise (the caller checks ``res is out``)."""
    global _ROOM_REQUEST, _ROOM_RESULT, _OUT_BUFFER
    if out is not None:
        _OUT_BUFFER = out
        try:
            return _feature_mlps(x, p, sum_features, return_total, out_dtype, total_rows, pad_ok, tables)
        finally:
            _OUT_BUFFER = None
    if not (room_rows and sum_features):
        return _feature_mlps(x, p, sum_features, return_total, out_dtype, total_rows, pad_ok, tables)
    _ROOM_REQUEST, _ROOM_RESULT = int(room_rows), None
    try:
        res = _feature_mlps(x, p, sum_features, return_total, out_dtype, total_rows, pad_ok, tables)
    finally:
        _ROOM_REQUEST = 0
    out = res[0] if return_total else res
    if _ROOM_RESULT is not None and _ROOM_RESULT[0] == out.data_ptr() and out.shape[0] + room_rows == _ROOM_RESULT[1].shape[0]:
        out.gnan_room = _ROOM_RESULT[1]
    _ROOM_RESULT = None
    return res


def _feature_mlps(x: torch.Tensor, p: StackedMLP, sum_features: bool, return_total: bool = False,
                  out_dtype=torch.float32, total_rows: Optional[int] = None, pad_ok: bool = False, tables=None):
    """``fx[n, k*C + c] = f_k(x[n, k])[c]`` or, with ``sum_features``, ``sum_k f_k(x[n, k])`` — GNAN.py:57-62,157.
    ``return_total`` additionally returns the column sums of the result (the aggregation's rest-bucket operand),
    fused into the look-up kernel where possible.  ``out_dtype=torch.bfloat16`` stores the per-feature rows in bf16
    (inference only; one output channel), the operand format of the bf16-storage aggregation.  ``total_rows`` limits
    the column sums to the first rows (a rank's owned rows ahead of its halo rows, ``distributed.halo_recompute_forward``).
    ``pad_ok``: the caller accepts extra all-zero feature columns in the per-feature result (large inputs with a ragged
    feature count are evaluated padded to a multiple of 16 features; without ``pad_ok`` the result is a strided view).
    ``tables``: the tables of ``p`` queued ahead of time (:class:`TablePrefetch`; inference only — ignored where the table
    path does not apply)."""
    _lib.require_device(x, p.w_last)
    if x.shape[1] != p.F:
        raise ValueError(f"x has {x.shape[1]} feature columns, the model was built for {p.F}")
    if tables is not None and not (PAD_FEATURES and p.F % PAD_FEATURES and p.C == 1):
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in p[:6]):
            raise _lib.GnanHipError("prefetched tables are an inference feature: run under torch.no_grad()")
        fused_total = return_total and total_rows != 0
        out, _, total = _fmlp_forward(x, p, sum_features, fused_total, False, out_dtype, total_rows, prebuilt=tables)
        if not return_total:
            return out
        if total_rows == 0:
            total = out.new_zeros(out.shape[1], dtype=torch.float32)
        if total is None:
            total = column_sums(out if total_rows is None else out[:total_rows])
        return out, total
    twin = getattr(p.w_last, "gnan_padded", None)           # modules.FlatMLPStore: the padded problem without the concatenations
    if (PAD_FEATURES and p.F % PAD_FEATURES and p.C == 1 and p.L >= 2
            and x.shape[0] * p.F >= (PAD_MIN_WORK_STORE if twin is not None else PAD_MIN_WORK)):
        # Real inputs have F = raw features + the ones column (129 for arxiv / papers100M): rows that are not 16-byte
        # aligned and a last feature group that is not whole, i.e. the general look-up kernel, scalar operand gathers
        # (F = 65 instead of 64 on the 10M-node graph: 22.4 instead of 5.7 ms per forward) and no bf16 rows.  The
        # problem is padded to a multiple of 16 features with all-zero shape functions instead: x once per (static)
        # feature matrix, the stacked weights per call (a few tiny concatenations autograd sees through).
        Fp = (p.F + PAD_FEATURES - 1) // PAD_FEATURES * PAD_FEATURES
        padded = twin if (twin is not None and twin.F == Fp) else _padded_stack(p, Fp)
        res = _feature_mlps(_padded_x(x, Fp), padded, sum_features, return_total, out_dtype, total_rows)
        if sum_features or pad_ok:              # [n, C] either way; or the caller takes the zero columns along
            return res
        if return_total:
            return res[0][:, :p.F], res[1][:p.F]
        return res[:, :p.F]
    return _FeatureMLPs.apply(x, sum_features, return_total, out_dtype, total_rows, p.L, p.H, p.C, p.F,
                              p.w_first, p.b_first, p.w_mid, p.b_mid, p.w_last, p.b_last)


# =============================================================================
# reductions over rows (csrc/colsum.hip)
# =============================================================================
def column_sums(S: torch.Tensor) -> torch.Tensor:
    """``total[w] = sum_j S[j, w]`` (``gnan_colsum``): the rest-bucket operand of the aggregation."""
    _lib.require_device(S)
    S = S.detach()
    bf16 = S.dtype == torch.bfloat16
    if not bf16:
        S = S.float()
    S = _rows(S)
    n, W = S.shape
    total = torch.empty(W, dtype=torch.float32, device=S.device)
    need = _lib.lib().gnan_colsum_workspace_bytes(W)
    ws = torch.empty(need // 8, dtype=torch.float64, device=S.device)
    fn = _lib.lib().gnan_colsum_bf16 if bf16 else _lib.lib().gnan_colsum
    _lib.check(fn(_lib.ptr(S), n, W, S.stride(0), _lib.ptr(total), _lib.ptr(ws), need, _lib.stream_of(S)),
               "gnan_colsum")
    return total


def column_sums_weighted(S: torch.Tensor, cnt_col: torch.Tensor, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``total[w] = scale * sum_r S[r, w] / max(cnt_col[r], 1)`` (``gnan_colsum_weighted``): ``cnt_col`` a column VIEW of the shell-count
    table (int32, any row stride), ``scale`` a device scalar — the rest bucket's pull on every operand row in the wide backward."""
    _lib.require_device(S, cnt_col)
    S = _rows(S.detach().float())
    n, W = S.shape
    if cnt_col.dtype != torch.int32 or cnt_col.dim() != 1 or cnt_col.shape[0] != n:
        raise ValueError("column_sums_weighted: one int32 count per row")
    total = torch.empty(W, dtype=torch.float32, device=S.device)
    need = _lib.lib().gnan_colsum_workspace_bytes(W)
    ws = torch.empty(need // 8, dtype=torch.float64, device=S.device)
    _lib.check(_lib.lib().gnan_colsum_weighted(_lib.ptr(S), n, W, S.stride(0), _lib.ptr(cnt_col), cnt_col.stride(0),
                                               _lib.ptr(scale), _lib.ptr(total), _lib.ptr(ws), need, _lib.stream_of(S)),
               "gnan_colsum_weighted")
    return total


def weight_table(lut: torch.Tensor, cnt: Optional[torch.Tensor], n: int, with_rest: bool) -> torch.Tensor:
    """``wt [n, D, Cw] = lut[d, c] / max(cnt[i, d], 1) - (with_rest ? lut[D-1, c] / max(cnt[i, D-1], 1) : 0)`` (``gnan_weight_table``):
    the per-node weights the wide backward reads by neighbour."""
    _lib.require_device(lut, cnt)
    lut = lut.detach().float().contiguous()
    D, Cw = lut.shape
    wt = torch.empty((n, D, Cw), dtype=torch.float32, device=lut.device)
    _lib.check(_lib.lib().gnan_weight_table(_lib.ptr(lut), _lib.ptr(cnt), 0 if cnt is None else cnt.stride(0), n, D, Cw, int(with_rest),
                                            _lib.ptr(wt), _lib.stream_of(lut)), "gnan_weight_table")
    return wt


def feature_sum(fx: torch.Tensor, C: int) -> torch.Tensor:
    """``out[i, c] = sum_k fx[i, k * C + c]`` (``gnan_feature_sum``; C in {1, 2, 4}, whole 16-byte quads per row)."""
    _lib.require_device(fx)
    fx = _rows(fx.detach().float())
    n, W = fx.shape
    out = torch.empty((n, C), dtype=torch.float32, device=fx.device)
    _lib.check(_lib.lib().gnan_feature_sum(_lib.ptr(fx), n, W, fx.stride(0), C, _lib.ptr(out), out.stride(0),
                                           _lib.stream_of(fx)), "gnan_feature_sum")
    return out


class _GraphReadout(torch.autograd.Function):
    """``out[c] = sum_i Y[i, c]`` — the graph read-out of GNAN.py:75-79 / models.py:383-384 — by ``gnan_colsum`` (float64
    across threads, fixed order); backward: every node receives the read-out's gradient (a broadcast view, no arithmetic)."""

    @staticmethod
    def forward(ctx, Y):
        ctx.n = Y.shape[0]
        return column_sums(Y).to(Y.dtype)

    @staticmethod
    def backward(ctx, d):
        return d.reshape(1, -1).expand(ctx.n, -1)


def graph_readout(Y: torch.Tensor) -> torch.Tensor:
    """``[N, C] -> [C, 1]`` (what ``forward`` of a graph task returns, GNAN.py:79)."""
    return _GraphReadout.apply(Y).view(-1, 1)
