"""Host-side mirror of the reference's model interface for the aggregation path.

Same class names, constructor signatures, ``forward`` signatures, parameter
registration order (hence the same RNG stream at construction) and
``state_dict`` keys as the reference:

* :class:`StandaloneTensorGNAN`, :class:`StandaloneGNAN` — the stand-alone model file
  (GNAN.py:9-79, GNAN.py:82-176); exported as ``gnan_amd.GNAN.TensorGNAN`` / ``.GNAN``;
* :class:`NAM`, :class:`TensorGNAN`, :class:`GNAN` — the copy ``main.py`` imports
  (models.py:259-300, 303-384, 387-481); exported from ``gnan_amd.models``.

``forward`` never evaluates rho per pair and never loops over features or nodes in
Python: it stacks the per-feature weights, launches the fused shape-function kernel,
evaluates rho on the handful of distinct distances, and launches the hop-coded
aggregation kernel (``functional.py`` -> ``libgnan_hip.so``).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import _lib
from ._cache import TensorKeyedCache
from .aggregate import rho_aggregate
from .functional import StackedMLP, feature_mlps, graph_readout
from .graph import HopGraph, hop_inputs


# =============================================================================
# construction helpers (layouts fixed by the reference's state_dict keys)
# =============================================================================
def _shape_mlp(n_layers: int, hidden: Optional[int], out: int, bias: bool, dropout: Optional[float]) -> nn.Sequential:
    """``R -> R^out`` MLP.  With ``dropout`` given: Linear/ReLU/Dropout triples (keys 0,3,6,… —
    GNAN.py:24-34); with ``dropout=None``: Linear/ReLU pairs (keys 0,2,4,… — GNAN.py:38-47)."""
    if n_layers == 1:
        return nn.Sequential(nn.Linear(1, out, bias=bias))
    mods, width_in = [], 1
    for _ in range(n_layers - 1):
        mods += [nn.Linear(width_in, hidden, bias=bias), nn.ReLU()]
        if dropout is not None:
            mods.append(nn.Dropout(p=dropout))
        width_in = hidden
    mods.append(nn.Linear(hidden, out, bias=bias))
    return nn.Sequential(*mods)


def _tiny_init(module: nn.Module) -> None:
    """The TensorGNAN re-initialisation: xavier-normal with gain 0.01, zero biases (GNAN.py:49-53)."""
    for name, p in module.named_parameters():
        if "weight" in name:
            nn.init.xavier_normal_(p, gain=0.01)
        elif "bias" in name:
            nn.init.constant_(p, 0)


GRAPH_CACHE_ENTRIES = 8192   # hop-coded graphs kept per model (graph tasks cycle through a few thousand small graphs per epoch)

# ``model.parameters()`` is torch's own: the F x L per-layer tensors, in ``named_parameters()`` order — the nn.Module contract
# (DDP hooks, tooling that zips or counts the two, optimizer state_dicts).  The flat buffers those tensors are views of are handed
# out by ``model.flat_parameters()`` (= ``gnan_amd.optim_params(model)``): an optimizer built over THEM zeroes and updates a dozen
# tensors per step instead of 8604 (Cora shape) and computes the same numbers (every element-wise optimizer does); give an
# optimizer one face or the other, never both.  True: ``parameters()`` itself yields the flat buffers, as in round 5 — an opt-in for
# an unchanged ``Adam(model.parameters())`` (main.py:141) that wants the cheap step and knows what it gives up.
FLAT_PARAMETERS = False
PAD_STORE_FEATURES = True    # FlatMLPStore keeps room for F rounded up to 16 features (all-zero shape functions): see rebuild()


class FlatMLPStore:
    """The F x L per-feature ``nn.Linear`` parameters of a ``ModuleList`` of identical MLPs, re-homed into six
    contiguous buffers (first / middle / last layer, weights and biases) of which every Parameter is a view.

    The Parameters keep their identity, names and shapes (``state_dict`` keys, optimizers, ``fs[k](x)`` all work as
    before), but the kernels read the stacked buffers directly: no ``torch.stack`` of 10^2..10^4 small tensors per
    forward, no un-stacking in backward.  Gradients reach the Parameters through six proxy leaves: a hook adds the
    stacked gradient into a flat gradient buffer of which every ``Parameter.grad`` is a view (re-linked only after
    ``zero_grad(set_to_none=True)`` dropped the views).  In-place updates (optimizer steps, ``load_state_dict``)
    keep the sharing; ``.to()`` / ``.float()`` re-home (:meth:`rebuild`)."""

    def __init__(self, mlps):
        self.mlps = mlps
        self.flat = {}                                 # name -> Parameter over buf[name] (same storage); identity survives rebuilds
        self.rebuild()

    def rebuild(self) -> None:
        lin = [[m for m in seq if isinstance(m, nn.Linear)] for seq in self.mlps]
        self.lin, self.F, self.L = lin, len(lin), len(lin[0])
        self.has_bias = lin[0][0].bias is not None
        L = self.L
        self.C = lin[0][-1].out_features
        self.H = lin[0][0].out_features if L >= 2 else 0

        # A ragged feature count (F = raw features + the ones column: 129 on ogbn-arxiv / papers100M) keeps the fast look-up and
        # moment kernels away (whole 16-feature groups, 16-byte rows).  The buffers are therefore allocated for Fp = F rounded
        # up to 16 features, the extra ones all-zero shape functions that no Parameter, gradient view or optimizer ever sees:
        # ``buf`` / ``grad`` / the flat Parameters are the first F features (contiguous: one layer per buffer), ``pbuf`` /
        # ``pgrad`` the whole padded storage the kernels may take instead (``stacked(...).w_last.gnan_padded``) — what
        # functional._padded_stack builds with six concatenations per forward, for free.
        pad = PAD_STORE_FEATURES and self.F >= 16 and self.F % 16 and self.C == 1 and L in (2, 3)
        self.Fp = (self.F + 15) // 16 * 16 if pad else self.F

        def home(layers, attr):                       # stack [len(layers), Fp, ...] and turn the Parameters into views
            with torch.no_grad():
                buf = torch.stack([torch.stack([getattr(lin[k][l], attr).data for k in range(self.F)], 0)
                                   for l in layers], 0).contiguous()
                if self.Fp != self.F:
                    whole = buf.new_zeros((buf.shape[0], self.Fp) + tuple(buf.shape[2:]))
                    whole[:, :self.F] = buf
                    buf = whole
                for i, l in enumerate(layers):
                    for k, view in enumerate(buf[i].unbind(0)[:self.F]):      # one C++ call for the views
                        getattr(lin[k][l], attr).data = view
            return buf
        first, mid, last = ([0], list(range(1, L - 1)), [L - 1]) if L >= 2 else ([], [], [0])
        self.slots = {"first": first, "mid": mid, "last": last}
        self.buf, self.pbuf = {}, {}
        for part, layers in self.slots.items():
            if layers:
                for attr, tag in (("weight", "_w"), ("bias", "_b")):
                    if tag == "_b" and not self.has_bias:
                        continue
                    whole = home(layers, attr)
                    self.pbuf[part + tag] = whole
                    self.buf[part + tag] = whole[:, :self.F] if self.Fp != self.F else whole
        self.grad, self.pgrad, self.grad_views, self.pending = {}, {}, {}, {}
        self.touched = set()                          # buffers a backward pass delivered a gradient for (replay.py reads and clears it)
        track = bool(lin[0][-1].weight.requires_grad)
        for name, buf in self.buf.items():
            fp = self.flat.get(name)
            if fp is None:
                self.flat[name] = nn.Parameter(buf, requires_grad=track)
            else:                                     # moved (.to() / .float()): an optimizer built over the flat Parameters keeps them
                fp.data = buf
                fp.grad = None

    def params(self):
        """Every per-layer Parameter that is a view of this store's buffers."""
        for layers in self.lin:
            for m in layers:
                yield m.weight
                if m.bias is not None:
                    yield m.bias

    def consistent(self) -> bool:
        """Cheap guard: the first and the last Parameter still live inside the buffers."""
        a, b = self.lin[0][-1].weight, self.lin[-1][-1].weight
        w = self.buf["last_w"]
        return a.data_ptr() == w[0, 0].data_ptr() and b.data_ptr() == w[0, self.F - 1].data_ptr() and a.device == w.device

    def _kernel_view(self, name: str, t: torch.Tensor) -> torch.Tensor:
        if name in ("first_w",):
            return t[0, ..., 0]                       # [1, F, H, 1] -> [F, H]
        if name in ("first_b", "last_b"):
            return t[0]
        if name == "last_w":
            return t[0, ..., 0] if self.L == 1 else t[0]     # L == 1: Linear(1, C) -> [F, C]
        return t                                      # mid_w [L-2, F, H, H], mid_b [L-2, F, H]

    def _link_grads(self, name: str) -> None:
        """Point every Parameter's ``.grad`` at its slice of the flat gradient buffer (the (param, view) pairs are
        built once per buffer; after ``zero_grad(set_to_none=True)`` only the F x L assignments are repeated)."""
        pairs = self.grad_views.get(name)
        if pairs is None:
            part, attr = name.split("_")
            g = self.grad[name]
            which = "weight" if attr == "w" else "bias"
            pairs = [(getattr(self.lin[k][l], which), view)
                     for i, l in enumerate(self.slots[part]) for k, view in enumerate(g[i].unbind(0))]
            self.grad_views[name] = pairs
        for param, view in pairs:
            param.grad = view

    def views_like(self, name: str, flat: torch.Tensor):
        """``[(Parameter, view of flat)]`` for a tensor shaped like ``buf[name]``: the slice of ``flat`` that lies where the
        Parameter lies in its buffer (optimizer state kept flat, ``graphed.FlatAdamStep``)."""
        part, attr = name.split("_")
        which = "weight" if attr == "w" else "bias"
        return [(getattr(self.lin[k][l], which), view)
                for i, l in enumerate(self.slots[part]) for k, view in enumerate(flat[i].unbind(0))]

    def _linked(self, name: str) -> bool:
        g = self.grad.get(name)
        if g is None:
            return False
        part, attr = name.split("_")
        p0 = getattr(self.lin[0][self.slots[part][0]], "weight" if attr == "w" else "bias")
        p1 = getattr(self.lin[-1][self.slots[part][-1]], "weight" if attr == "w" else "bias")
        return (p0.grad is not None and p1.grad is not None and p0.grad.data_ptr() == g[0, 0].data_ptr()
                and p1.grad.data_ptr() == g[-1, -1].data_ptr())

    def _occupied(self, name: str) -> bool:
        """Does the flat gradient buffer hold gradients a new one must be ADDED to?  Not after either kind of caller zeroed
        them: ``zero_grad()`` of an optimizer over ``model.parameters()`` drops the flat Parameter's ``.grad`` (the per-layer
        views stay linked to the buffer — nothing of size F x L happens per step), one over the per-layer Parameters drops
        theirs.  (``set_to_none=False`` zeroes the shared buffer in place, and adding to zeros is right.)"""
        return self._linked(name) and self.flat[name].grad is not None

    def grad_dest(self, name: str, shape, device) -> Optional[torch.Tensor]:
        """Where a gradient kernel may write the stacked gradient of ``buf[name]`` directly — the flat gradient buffer, in the
        kernel's view of it — or None when the buffer holds gradients that must be added to (``.grad`` already linked: a
        second backward pass before ``zero_grad``) or has been handed out already.  The tensor comes back through autograd as
        the proxy leaf's gradient; :meth:`_on_grad` recognises it and only links the views (no copy: 5 us per buffer, twelve
        per step of a small graph)."""
        if self.pending.get(name) is not None or self._occupied(name) or self.buf[name].dtype != torch.float32:
            return None
        self._grad_buffer(name)
        view = self._kernel_view(name, self.grad[name])
        if tuple(view.shape) != tuple(shape) and self.Fp != self.F:
            view = self._kernel_view(name, self.pgrad[name])      # the padded twin's gradient: the whole padded buffer
        if view is self.grad[name] or view is self.pgrad[name]:
            view = view.view(view.shape)              # a tensor object of its own: autograd keeps (not clones) a gradient nobody else holds
        if view.shape != shape or view.device != device or not view.is_contiguous():
            return None
        self.pending[name] = view
        return view

    def _grad_buffer(self, name: str) -> torch.Tensor:
        """The persistent gradient buffer of ``buf[name]`` (first F features of ``pgrad[name]``, which mirrors ``pbuf[name]``)."""
        if name not in self.grad:
            whole = torch.zeros_like(self.pbuf[name])
            self.pgrad[name] = whole
            self.grad[name] = whole[:, :self.F] if self.Fp != self.F else whole
        return self.grad[name]

    def _on_grad(self, name: str, g: torch.Tensor) -> None:
        self.touched.add(name)
        handed = self.pending.get(name)
        if handed is not None and g.data_ptr() == handed.data_ptr() and g.numel() == handed.numel():
            self.pending[name] = None                 # written in place by the kernel (grad_dest)
        else:
            self._grad_buffer(name)
            # (a gradient of the padded twin has the padded buffer's size; its extra features' entries are nobody's gradient)
            dest = self.pgrad[name] if (self.Fp != self.F and g.numel() == self.pgrad[name].numel()) else self.grad[name]
            g = g.reshape(dest.shape).to(dest.dtype)
            if self._occupied(name) or handed is not None:
                dest.add_(g)                          # ordinary autograd accumulation, on the flat buffer (a pending direct
            else:                                     # write already sits in it)
                dest.copy_(g)                         # the buffer (and the views cut from it) persists across steps
        if not self._linked(name):
            self._link_grads(name)
        fp = self.flat[name]
        if fp.grad is not self.grad[name]:
            fp.grad = self.grad[name]

    def direct_use(self) -> None:
        """Call before the per-layer modules themselves (``self.mlps[k](x)``) take part in an autograd graph: autograd then
        accumulates straight into the per-layer ``.grad`` tensors, so these must be views of the flat gradient buffers and the
        flat Parameters must show those buffers — zeroed first if the last ``zero_grad`` dropped them."""
        if not (torch.is_grad_enabled() and self.lin[0][-1].weight.requires_grad):
            return
        for name, buf in self.buf.items():
            if name not in self.grad:
                self._grad_buffer(name)
            elif not self._occupied(name):
                self.grad[name].zero_()
            self.pending[name] = None
            if not self._linked(name):
                self._link_grads(name)
            if self.flat[name].grad is not self.grad[name]:
                self.flat[name].grad = self.grad[name]

    def stacked(self, track_grad: bool) -> StackedMLP:
        if track_grad and any(v is not None for v in self.pending.values()):
            self.pending.clear()                      # a backward pass that claimed a destination never finished

        def views(bufs, F):
            out = {}
            for name, t in bufs.items():
                v = self._kernel_view(name, t)
                if track_grad:
                    v = v.detach().requires_grad_(True)
                    v.register_hook(lambda g, name=name: self._on_grad(name, g))
                    v.gnan_grad_dest = lambda shape, device, name=name: self.grad_dest(name, shape, device)
                out[name] = v
            return StackedMLP(out.get("first_w"), out.get("first_b"), out.get("mid_w"), out.get("mid_b"),
                              out["last_w"], out.get("last_b"), self.L, self.H, self.C, F)
        st = views(self.buf, self.F)
        if self.Fp != self.F:
            # the same shape functions followed by Fp - F all-zero ones, over the padded storage: what the fast kernels take for
            # a ragged feature count (functional._feature_mlps); only ONE of the two sets of leaves receives a forward's gradient
            st.w_last.gnan_padded = views(self.pbuf, self.Fp)
        return st


class _PathBase(nn.Module):
    """Shared plumbing: flat parameter stores, hop-graph lookup, rho look-up tables."""

    def _init_caches(self):
        self._stores = {}
        # hop-coded graphs derived from the inputs' tensors, found again by the identity of those tensor objects (not by
        # their addresses: the allocator recycles those between the graphs of a batch_size=1 loop, trainer.py:46)
        self._graph_cache = TensorKeyedCache(GRAPH_CACHE_ENTRIES)

    def _mlp_groups(self):
        """``(store name, MLPs)`` of the per-feature networks this module owns (what ``_stacked`` is called with)."""
        groups = []
        if isinstance(getattr(self, "fs", None), nn.ModuleList):
            groups.append(("fs", self.fs))
        if isinstance(getattr(self, "rho", None), nn.Sequential):
            groups.append(("rho", [self.rho]))
        return groups

    def _ensure_stores(self) -> None:
        for name, mlps in self._mlp_groups():
            store = self._stores.get(name)
            if store is None:
                self._stores[name] = FlatMLPStore(mlps)
            elif not store.consistent():
                store.rebuild()

    def parameters(self, recurse: bool = True):
        """torch's own ``parameters()`` (the per-layer tensors) unless ``modules.FLAT_PARAMETERS`` is switched on."""
        if FLAT_PARAMETERS and recurse and "_stores" in self.__dict__:
            yield from self.flat_parameters()
        else:
            yield from super().parameters(recurse)

    def flat_parameters(self):
        """The flat buffers every per-layer Parameter is a view of, followed by whatever Parameter lies in none of them
        (``rhos[0 .. F-2]`` of ``GNAN(rho_per_feature=True)``, GNAN.py:108-123: tensors no forward reads): what to hand an
        optimizer INSTEAD of ``parameters()`` — same elements (``sum(p.numel())`` agrees), a dozen tensors.  An optimizer state
        saved over one face does not load into an optimizer over the other."""
        flats, managed = [], set()
        for m in self.modules():
            if isinstance(m, _PathBase) and "_stores" in m.__dict__:
                m._ensure_stores()
                for store in m._stores.values():
                    flats.extend(store.flat[name] for name in store.buf)
                    managed.update(id(p) for p in store.params())
        yield from flats
        for p in nn.Module.parameters(self, recurse=True):
            if id(p) not in managed:
                yield p

    def zero_grad(self, set_to_none: bool = True) -> None:
        """``Module.zero_grad`` for BOTH faces of the parameters: the flat Parameters (``parameters()``) and the per-layer views
        whose ``.grad`` are views of the same gradient buffers.  (An optimizer's own ``zero_grad()`` only knows the Parameters
        it was given: over ``model.parameters()`` it drops the flat gradients and leaves the per-layer ``.grad`` views standing
        — stale until the next backward pass rewrites the buffer they look into; nothing of size F x L happens per step.)"""
        super().zero_grad(set_to_none)
        for m in self.modules():
            for store in getattr(m, "_stores", {}).values():
                store.pending.clear()
                for name, g in store.grad.items():
                    if not set_to_none:
                        g.zero_()
                    elif store._linked(name):
                        for p, _ in store.grad_views.get(name, ()):
                            p.grad = None

    def requires_grad_(self, requires_grad: bool = True):
        for _, p in self.named_parameters():
            p.requires_grad_(requires_grad)
        for m in self.modules():
            for store in getattr(m, "_stores", {}).values():
                for fp in store.flat.values():
                    fp.requires_grad_(requires_grad)
        return self

    def _apply(self, fn, *args, **kwargs):               # .to() / .cuda() / .float(): every Parameter moved on its own
        out = super()._apply(fn, *args, **kwargs)
        for store in getattr(self, "_stores", {}).values():
            store.rebuild()                              # re-home them into contiguous buffers on the new device
        return out

    # ---- parameters -> stacked device tensors --------------------------------------------
    def _stacked(self, name: str, mlps) -> StackedMLP:
        """Stacked weights of the F per-feature MLPs — views of the :class:`FlatMLPStore`, no copies.  With autograd on,
        proxy leaves carry the stacked gradients back to the individual Parameters."""
        store = self._stores.get(name)
        if store is None:
            store = self._stores[name] = FlatMLPStore(mlps)
        elif not store.consistent():
            store.rebuild()
        return store.stacked(torch.is_grad_enabled() and store.lin[0][0].weight.requires_grad)

    def _check_dropout(self):
        """Training-mode Dropout (GNAN.py:28,32; run.sh trains with 0.6) runs in the kernels: ``gnan_fmlp_fwd`` /
        ``gnan_fmlp_bwd`` apply a keep-mask that is a counter-based hash of (seed, node, feature, layer, unit), the seed
        drawn from torch's generator once per forward (``functional.feature_mlps_dropout``).  Same distribution as
        ``nn.Dropout``, not torch's Bernoulli stream — which depends on the order of the reference's Python loop and is not
        part of any contract.  The table path cannot hold per-node masks, so these steps use the direct kernels (the
        reference only trains with Dropout in its first epoch: it never leaves eval mode again after ``trainer.py:97``);
        a one-time note says so."""
        if self._dropout_active() and not getattr(self, "_dropout_noted", False):
            import warnings
            warnings.warn("gnan_amd: training-mode Dropout (p=%g) is active: the shape functions run through the direct "
                          "kernels with hash-generated masks (not the table look-up) until .eval() is called" % self.dropout)
            self._dropout_noted = True

    def _dropout_active(self) -> bool:
        return bool(self.training and self.dropout and self.dropout > 0)

    def _features(self, x, name, mlps, sum_features: bool, pad_ok: bool = False, want_total: bool = False,
                  out_dtype=torch.float32, room_rows: int = 0):
        """Shape functions of all features: fused HIP kernels, or the Dropout cold path described above.
        ``pad_ok``: the per-feature result may carry extra all-zero columns (``functional.feature_mlps``).
        ``want_total``: returns ``(rows, column sums or None)`` — the rest bucket's operand comes out of the look-up pass
        instead of a second pass over the rows (``rho_aggregate(s_total=...)`` accounts for it in its backward)."""
        if self._dropout_active():
            from .functional import feature_mlps_dropout
            return feature_mlps_dropout(x, self._stacked(name, mlps), sum_features, float(self.dropout), return_total=want_total)
        kw = {} if out_dtype == torch.float32 else {"out_dtype": out_dtype}
        if room_rows:
            kw["room_rows"] = room_rows
        return feature_mlps(x, self._stacked(name, mlps), sum_features=sum_features, pad_ok=pad_ok,
                            return_total=want_total, **kw)

    def _operand(self, x, name, mlps, sum_features: bool, with_total: bool, pad_ok: bool = False, out_dtype=torch.float32,
                 graph: Optional[HopGraph] = None):
        """``(rows, column sums)`` of the shape functions — the sums are ``None`` unless ``with_total`` (a CSR's rest bucket
        needs them; a dense adjacency lists every pair).  ``graph``: the graph the rows will be aggregated over — a large
        CSR wants room behind narrow rows for the compact copy of its most listed nodes' rows (``HopGraph.hot_columns``)."""
        room = 0
        if sum_features and graph is not None and not graph.is_dense and not self._dropout_active():
            from .graph import HOT_COLUMNS, HOT_COLUMNS_MIN_NNZ
            room = HOT_COLUMNS if graph.nnz >= HOT_COLUMNS_MIN_NNZ else 0
        if not with_total:
            return self._features(x, name, mlps, sum_features, pad_ok=pad_ok, out_dtype=out_dtype, room_rows=room), None
        return self._features(x, name, mlps, sum_features, pad_ok=pad_ok, want_total=True, out_dtype=out_dtype, room_rows=room)

    def _mark(self, name: str) -> None:
        """Stage boundary of a forward (``start`` / ``lut`` / ``fmlp`` / ``spmm``): ``stage_hook`` — unset by default — is
        what bench.py records its HIP events through."""
        hook = getattr(self, "stage_hook", None)
        if hook is not None:
            hook(name)

    # ---- inputs -> hop-coded adjacency -----------------------------------------------------
    def _graph(self, inputs, want_norm: bool) -> HopGraph:
        from .functional import _pin_for_capture
        return _pin_for_capture(self._graph_lookup(inputs, want_norm))    # a captured step keeps its graph alive

    def _graph_lookup(self, inputs, want_norm: bool) -> HopGraph:
        g = getattr(inputs, "gnan_graph", None)
        if g is not None:
            return g
        if hasattr(inputs, "gnan_rowptr"):
            cnt = getattr(inputs, "gnan_cnt", None)
            src = (inputs.gnan_rowptr, inputs.gnan_col, inputs.gnan_code, cnt)
            extra = ("csr", int(inputs.x.shape[0]), int(inputs.gnan_n_codes))
            g = self._graph_cache.get(src, extra)
            if g is None:
                g = self._graph_cache.put(src, extra, HopGraph.from_csr(
                    inputs.gnan_rowptr, inputs.gnan_col, inputs.gnan_code, n_cols=inputs.x.shape[0],
                    n_codes=int(inputs.gnan_n_codes), cnt=cnt))
            return g
        nd = inputs.node_distances
        norm = inputs.normalization_matrix if want_norm else getattr(inputs, "normalization_matrix", None)
        g = self._graph_cache.get((nd, norm), "dense")
        if g is None:
            g = self._graph_cache.put((nd, norm), "dense", HopGraph.from_dense(nd, norm))
        return g

    def hop_graph(self, inputs) -> HopGraph:
        """The hop-coded graph ``forward(inputs)`` aggregates over (built from the dense inputs on first sight, then cached)."""
        want = True if isinstance(self, _GNANCore) else bool(getattr(self, "normalize_rho", True))
        return self._graph(inputs, want_norm=want)

    def _small_graph(self, x, g: HopGraph, use_cnt, graph_sum: bool):
        """The whole forward by one launch where ``gnan_small_graph_fwd`` applies (a small dense-coded graph, features summed
        per node: what a graph-level task feeds per step; ``use_cnt``: False, True = post-rho normalisation, "pre" =
        GNAN.py:65-67), else None."""
        from .small_graph import (SMALL_GRAPH_MAX_NODES, SlotGraph, slot_graph_applies, slot_graph_forward, small_graph_applies,
                                  small_graph_forward)
        if isinstance(g, SlotGraph):
            # a captured step of a batch-size-1 loop: the graph sits in slots, its size is the device's to know
            f, rho = self._stacked("fs", self.fs), self._stacked("rho", [self.rho])
            if use_cnt == "pre" or not graph_sum or not slot_graph_applies(g, f, rho):
                raise _lib.GnanHipError("graph slots serve graph-level read-outs with post-rho (or no) normalisation, a one-channel "
                                        "rho and at most 8 output channels")
            return slot_graph_forward(g, f, rho, bool(use_cnt))
        if not g.is_dense or x.shape[0] > SMALL_GRAPH_MAX_NODES:
            return None
        f, rho = self._stacked("fs", self.fs), self._stacked("rho", [self.rho])
        if not small_graph_applies(x, g, f, rho, pre_rho=use_cnt == "pre"):
            return None
        out = small_graph_forward(x, g, f, rho, use_cnt, graph_sum)
        for name in ("lut", "fmlp", "spmm"):
            self._mark(name)
        return out

    # ---- rho on the distinct distances -------------------------------------------------------
    def _lut_global(self, g: HopGraph) -> torch.Tensor:
        """``lut[d] = rho(float32(1/(1+d)))``, ``lut[D-1] = rho(0)`` — D rows instead of N^2 (models.py:368).  One launch of
        the shape-function kernel on rho's stacked layers (and one of ``gnan_fmlp_bwd`` in the backward pass) instead of the
        six to ten framework launches of ``self.rho(...)`` — 0.04 ms of a 2-ms forward on the 10M-node graph."""
        return feature_mlps(hop_inputs(g.n_codes, g.device).view(-1, 1), self._stacked("rho", [self.rho]), sum_features=False)

    def _lut_pre_rho(self, g: HopGraph) -> torch.Tensor:
        """``lut[i, d] = rho(u_d / cnt[i, d])`` — the pre-rho normalisation of GNAN.py:65-67, per shell: rho's exact
        piecewise-linear table looked up D times per row (``gnan_rho_row_lut``), or, for small graphs, the shape-function
        kernels on the N*D arguments.  No torch MLP pass, nothing of size N x H."""
        from .functional import rho_row_lut
        return rho_row_lut(g.cnt, hop_inputs(g.n_codes, g.device), self._stacked("rho", [self.rho]))


# =============================================================================
# the stand-alone model file (GNAN.py)
# =============================================================================
class StandaloneTensorGNAN(_PathBase):
    """``TensorGNAN`` of the stand-alone model file — constructor GNAN.py:10-53, forward GNAN.py:55-79.

    Pre-rho normalisation; rho always has ``out_channels`` outputs; ``rho_per_feature`` and
    ``readout_n_layers`` are accepted and ignored, as upstream.
    """

    def __init__(self, in_channels, out_channels, n_layers, hidden_channels=None, bias=True, dropout=0.0,
                 device='cpu', rho_per_feature=False, normalize_rho=True, is_graph_task=False, readout_n_layers=1):
        super().__init__()
        self.device = device
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.n_layers = n_layers
        self.bias = bias
        self.dropout = dropout
        self.rho_per_feature = rho_per_feature
        self.normalize_rho = normalize_rho
        self.is_graph_task = is_graph_task
        self.fs = nn.ModuleList(_shape_mlp(n_layers, hidden_channels, out_channels, bias, dropout)
                                for _ in range(in_channels))
        self.rho = _shape_mlp(n_layers, hidden_channels, out_channels, not is_graph_task, None)
        _tiny_init(self)
        self._init_caches()

    def forward(self, inputs):
        from . import cpu_route, replay
        if cpu_route.applies(self, inputs.x):         # a CPU module on CPU inputs (the reference's default device): plain torch
            return cpu_route.forward_standalone_tensor(self, inputs)
        return replay.run(self, inputs)               # the launches below, or — third call on the same inputs — their hipGraphs

    def _forward(self, inputs):
        self._check_dropout()
        x = inputs.x
        _lib.require_device(x)
        self._mark("start")
        g = self._graph(inputs, want_norm=bool(self.normalize_rho))
        if not self._dropout_active():
            small = self._small_graph(x, g, "pre" if self.normalize_rho else False, bool(self.is_graph_task))
            if small is not None:
                return small
        lut = None if self.normalize_rho else self._lut_global(g)
        self._mark("lut")
        S, total = self._operand(x, "fs", self.fs, True, not g.is_dense, graph=g)    # [N, C]
        self._mark("fmlp")
        if self.normalize_rho:
            # GNAN.py:65-67: rho(node_distances / normalization_matrix) — rho's exact table, D look-ups per row, in the
            # order the aggregation walks the rows (aggregate.pre_rho_aggregate)
            from .aggregate import pre_rho_aggregate
            Y = pre_rho_aggregate(g, S, self._stacked("rho", [self.rho]), hop_inputs(g.n_codes, g.device), s_total=total)
        else:
            Y = rho_aggregate(g, S, lut, use_cnt=False, s_total=total)                # [N, C]
        self._mark("spmm")
        if not self.is_graph_task:
            return Y                                                                  # GNAN.py:72-73,79
        return graph_readout(Y)                                                       # [C, 1]  GNAN.py:75-79


class _GNANCore(_PathBase):
    """``GNAN`` — constructor GNAN.py:83-137 / models.py:388-442, forward GNAN.py:146-172."""

    def _build(self, in_channels, out_channels, n_layers, hidden_channels, bias, dropout, device,
               normalize_rho, rho_per_feature):
        self.device = device
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.num_layers = n_layers
        self.bias = bias
        self.dropout = dropout
        self.rho_per_feature = rho_per_feature
        self.normalize_rho = normalize_rho
        self.fs = nn.ModuleList(_shape_mlp(n_layers, hidden_channels, out_channels, bias, dropout)
                                for _ in range(in_channels))
        # rho has one output unless rho_per_feature (then one per output channel, GNAN.py:118-121);
        # a single-layer rho is always out_channels wide (GNAN.py:125-126).
        width = out_channels if (rho_per_feature or n_layers == 1) else 1
        if rho_per_feature:
            # upstream builds F copies and keeps using the LAST one under the name `rho`
            # (GNAN.py:108-123,137): `rho` shares its Linear modules with `rhos[F-1]`.
            self.rhos = nn.ModuleList(_shape_mlp(n_layers, hidden_channels, width, bias, None)
                                      for _ in range(in_channels))
            self.rho = nn.Sequential(*list(self.rhos[-1]))
        else:
            self.rho = _shape_mlp(n_layers, hidden_channels, width, bias, None)
        self._init_caches()

    def forward(self, inputs, node_ids=None):
        from . import cpu_route, replay
        if cpu_route.applies(self, inputs.x):
            return cpu_route.forward_gnan(self, inputs, node_ids)
        return replay.run(self, inputs, node_ids)

    def _forward(self, inputs, node_ids=None):
        self._check_dropout()
        x = inputs.x
        _lib.require_device(x)
        self._mark("start")
        g = self._graph(inputs, want_norm=True)            # GNAN.py:161 reads it unconditionally
        if node_ids is None and not self._dropout_active():
            small = self._small_graph(x, g, bool(self.normalize_rho), False)
            if small is not None:
                return small
        lut = self._lut_global(g)
        self._mark("lut")
        S, total = self._operand(x, "fs", self.fs, True, not g.is_dense, graph=g)    # f_sums, GNAN.py:157
        self._mark("fmlp")
        rows = None
        if node_ids is not None:
            rows = torch.as_tensor(list(node_ids) if not torch.is_tensor(node_ids) else node_ids,
                                   dtype=torch.int32, device=x.device)
        Y = rho_aggregate(g, S, lut, use_cnt=bool(self.normalize_rho), row_ids=rows, s_total=total)
        self._mark("spmm")
        return Y

    def print_rho_params(self):
        for name, param in self.rho.named_parameters():
            print(name, param)


class StandaloneGNAN(_GNANCore):
    def __init__(self, in_channels, out_channels, n_layers, hidden_channels=None, bias=True, dropout=0.0,
                 device='cpu', normalize_rho=True, rho_per_feature=False):
        super().__init__()
        self._build(in_channels, out_channels, n_layers, hidden_channels, bias, dropout, device,
                    normalize_rho, rho_per_feature)


# =============================================================================
# the copy main.py imports (models.py)
# =============================================================================
class NAM(_PathBase):
    """``NAM`` — models.py:259-300: ``sum_k f_k(x[:, k])``."""

    def __init__(self, in_channels, out_channels, num_layers, hidden_channels=None, bias=True, dropout=0.0,
                 device='cpu'):
        super().__init__()
        self.device = device
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self.bias = bias
        self.dropout = dropout
        self.fs = nn.ModuleList(_shape_mlp(num_layers, hidden_channels, out_channels, bias, dropout)
                                for _ in range(in_channels))
        self._init_caches()

    def forward(self, x):
        from . import cpu_route
        if cpu_route.applies(self, x):
            return cpu_route.forward_nam(self, x)
        self._check_dropout()
        return self._features(x, "fs", self.fs, True)


class TensorGNAN(_PathBase):
    """``TensorGNAN`` as ``main.py`` uses it — constructor models.py:304-356, forward models.py:358-384.

    Post-rho normalisation; rho is one-wide unless ``rho_per_feature``; graph tasks may end in a
    ``NAM`` read-out over the per-feature aggregates.
    """

    def __init__(self, in_channels, out_channels, n_layers, hidden_channels=None, bias=True, dropout=0.0,
                 device='cpu', rho_per_feature=False, normalize_rho=True, is_graph_task=False, readout_n_layers=1):
        super().__init__()
        self.device = device
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.n_layers = n_layers
        self.bias = bias
        self.dropout = dropout
        self.rho_per_feature = rho_per_feature
        self.normalize_rho = normalize_rho
        self.is_graph_task = is_graph_task
        self.readout_n_layers = readout_n_layers
        self.aggregation_order = "sum_first"          # or "reference" (models.py:373-376 evaluation order)
        self.operand_dtype = torch.float32            # or torch.bfloat16: storage of the reference order's [N, F*C] rows (inference)
        with_readout = bool(is_graph_task and readout_n_layers > 0)
        self.actual_output_dim_f = 1 if with_readout else out_channels                       # models.py:320
        self.actual_output_dim_rho = 1 if (not rho_per_feature or with_readout) else out_channels  # :321
        self.fs = nn.ModuleList(_shape_mlp(n_layers, hidden_channels, self.actual_output_dim_f, bias, dropout)
                                for _ in range(in_channels))
        self.rho = _shape_mlp(n_layers, hidden_channels, self.actual_output_dim_rho, not is_graph_task, None)
        if with_readout:
            self.readout_nam = NAM(in_channels, out_channels, readout_n_layers, hidden_channels, bias, dropout,
                                   device)
        _tiny_init(self)
        self._init_caches()

    def forward(self, inputs):
        from . import cpu_route, replay
        if cpu_route.applies(self, inputs.x):
            return cpu_route.forward_models_tensor(self, inputs)
        return replay.run(self, inputs)

    def _forward(self, inputs):
        self._check_dropout()
        x = inputs.x
        _lib.require_device(x)
        self._mark("start")
        g = self._graph(inputs, want_norm=bool(self.normalize_rho))
        with_readout = self.is_graph_task and self.readout_n_layers > 0
        if not with_readout and self.aggregation_order != "reference" and not self._dropout_active():
            small = self._small_graph(x, g, bool(self.normalize_rho), bool(self.is_graph_task))
            if small is not None:
                return small
        use_cnt = bool(self.normalize_rho)
        rest = not g.is_dense                 # a CSR lists some pairs only: the others weigh rho(0) on the column sums
        if with_readout:
            if not self._dropout_active() and not self.readout_nam._dropout_active() and g.is_dense:
                # a small graph: shape functions, rho, aggregation and the NAM read-out in ONE launch (csrc/small_graph_nam.hip)
                from .small_graph import small_graph_nam_applies, small_graph_nam_forward
                f, rho = self._stacked("fs", self.fs), self._stacked("rho", [self.rho])
                nam = self.readout_nam._stacked("fs", self.readout_nam.fs)
                if small_graph_nam_applies(x, g, f, rho, nam):
                    out = small_graph_nam_forward(x, g, f, rho, nam, use_cnt)
                    for name in ("fmlp", "spmm"):
                        self._mark(name)
                    return out
            lut = self._lut_global(g)
            self._mark("lut")
            fx, total = self._operand(x, "fs", self.fs, False, rest, pad_ok=True)     # [N, F]   (f is 1-wide; + zero columns)
            self._mark("fmlp")
            hidden = graph_readout(rho_aggregate(g, fx, lut, use_cnt, s_total=total)).view(1, -1)[:, :x.shape[1]]   # [1, F]   models.py:379 (gnan_colsum)
            self._mark("spmm")
            return self.readout_nam(hidden).T                                         # [C, 1]   models.py:380-384
        lut = self._lut_global(g)
        self._mark("lut")
        if self.aggregation_order == "reference":
            # the upstream evaluation order (models.py:373-376): aggregate every feature column, then sum
            # over features.  Same function, F times the aggregation traffic; kept because the intermediate
            # is the per-feature contribution tensor mf[c, i, k] and because BASELINE's workload is stated
            # in this order.
            from .aggregate import reference_order_applies, reference_order_forward
            stacked = None if self._dropout_active() or self.operand_dtype != torch.float32 else self._stacked("fs", self.fs)
            if stacked is not None and reference_order_applies(x, stacked, lut, g):
                # training at scale: one node whose backward pass is the sum-first order's (the read-out makes every
                # feature's row gradient the same [N, C] vector)
                Y = reference_order_forward(g, x, stacked, lut, use_cnt)
                self._mark("fmlp")
                self._mark("spmm")
                return Y if not self.is_graph_task else graph_readout(Y)
            fx, total = self._operand(x, "fs", self.fs, False, rest, pad_ok=True,
                                      out_dtype=self.operand_dtype)                   # [N, F*C] (+ zero columns when C == 1)
            self._mark("fmlp")
            Y = rho_aggregate(g, fx, lut, use_cnt, s_total=total, reduce_channels=self.actual_output_dim_f)   # [N, C]
        else:
            S, total = self._operand(x, "fs", self.fs, True, rest, graph=g)           # [N, C]  sum-first
            self._mark("fmlp")
            Y = rho_aggregate(g, S, lut, use_cnt, s_total=total)                      # [N, C]
        self._mark("spmm")
        if not self.is_graph_task:
            return Y                                                                  # models.py:375-376,384
        return graph_readout(Y)                                                       # [C, 1]  models.py:383-384

    def feature_contributions(self, inputs, _g=None, _lut=None):
        """``mf[i, k, c] = sum_j m_ij[c] * f_k(x[j, k])[c]`` — the per-feature aggregate of models.py:373,
        returned as ``[N, F, C]`` (the reference holds it as ``[C, N, F]``)."""
        x = inputs.x
        g = self._graph(inputs, want_norm=bool(self.normalize_rho)) if _g is None else _g
        lut = self._lut_global(g) if _lut is None else _lut
        fx = self._features(x, "fs", self.fs, False, pad_ok=True)                     # [N, F*C] (+ zero columns when C == 1)
        Y = rho_aggregate(g, fx, lut, bool(self.normalize_rho))                       # [N, F*C]
        return Y.view(x.shape[0], -1, self.actual_output_dim_f)[:, :x.shape[1]]


class GNAN(_GNANCore):
    """``models.GNAN``: identical to the stand-alone one except that the depth argument is spelled
    ``num_layers`` (models.py:388; ``main.py:79-83`` passes it by keyword).  Both spellings work."""

    def __init__(self, in_channels, out_channels, num_layers=None, hidden_channels=None, bias=True, dropout=0.0,
                 device='cpu', normalize_rho=True, rho_per_feature=False, n_layers=None):
        super().__init__()
        depth = num_layers if num_layers is not None else n_layers
        if depth is None:
            raise TypeError("GNAN() missing the depth argument (num_layers / n_layers)")
        self._build(in_channels, out_channels, depth, hidden_channels, bias, dropout, device,
                    normalize_rho, rho_per_feature)
