"""``model.forward(data)`` and the backward pass behind it as TWO hipGraph replays, inside the reference's UNCHANGED loop.

``harness.train_epoch`` replays a whole step — forward, loss, backward, optimizer — from one hipGraph, but only for callers
who swap ``trainer.py`` for it.  Under the reference's own loop (trainer.py:23-86: ``zero_grad``, ``model.forward(data)``, the
mask, the loss, ``loss.backward()``, ``optimizer.step()``, all of it eager and inside anomaly mode) the loss and the update are
the CALLER's code; what this package owns is the forward and the backward pass of its modules, and on a full-batch node task
those run on the same tensors every epoch (main.py builds one ``Data`` object and trains on it for hundreds of epochs).

So the modules capture themselves: the third ``forward`` on the same input tensors (object identity + version, as every
cache here) captures the forward into one hipGraph and — with autograd on — the backward pass of the autograd graph that
forward built into a second one.  From then on ``forward`` is one graph launch that returns (a copy of) its static output behind a
single autograd node, and ``loss.backward()`` reaches that node, which copies the incoming gradient into a static buffer and
launches the second graph; the parameter gradients land where they always land (the flat gradient buffers of
``modules.FlatMLPStore``), so ``optimizer.step()`` and ``named_parameters()`` see nothing new.  ~25 + ~45 launches and their
Python on the arxiv shape become two.

What a plan freezes and how it is kept honest: the input tensors (key), the parameters' addresses, train / eval mode, the
modules' switches (key); the sizes of the shape-function tables — the captured look-up checks the tables its own build
produced (``gnan_pwl_check_fit`` -> the plan's guard flag), the flag is read right after the forward replay, and a tripped plan
is dropped and the forward runs eagerly (the look-up then sizes itself); a second forward replay before the backward of the
first would overwrite what that backward reads — refused loudly.  Not captured: training-mode Dropout, ``node_ids``,
``x.requires_grad``, anything while a capture is already running (``harness`` capturing a whole step).  Anomaly mode is switched off DURING the capture
(its NaN checks read the device) and nowhere else.

Graph-level tasks hand over another small graph every step (batch_size = 1), so there is no input to key a plan on.  Those
forwards go through GRAPH SLOTS instead (``small_graph.SlotGraph``: static buffers whose kernels read the graph's size from the
device): one plan per (node tier, hop-code tier) is captured over the slots, and every graph that fits is copied in — one
launch — and replayed, forward and backward one launch each (``_run_small``).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

REPLAY_FORWARD = True        # off: every forward issues its launches eagerly
REPLAY_AFTER = 2             # eager forwards on the same inputs before the capture (they are its warm-up: caches, table sizes)
REPLAY_MIN_NODES = 129       # below: the small-graph route (graph slots, `_run_small`)
REPLAY_PLANS = 24            # plans a model keeps (train / eval x a few inputs; the graphs of a graph-level task that are too large for
                             # the slots — 129 to 417 nodes on Mutagenicity — get one each).  A full book takes no new inputs (they stay
                             # eager) rather than drop a live plan for them: a loop over more inputs than plans would otherwise capture,
                             # evict and capture again for ever
REPLAY_SMALL_GRAPHS = True   # graph-level forwards of small dense graphs through graph slots (one plan per tier, any graph)
REPLAY_COPY_MAX_BYTES = None        # None (default): a replayed forward ALWAYS hands out a copy of its static output — what an eager
                                    # forward returns is the caller's to keep (last epoch's logits, a collected evaluation output), so
                                    # the next replay must not rewrite it.  An int: outputs above that many bytes are the static tensor
                                    # ITSELF, valid until the next forward on the same inputs (an opt-in that saves the copy: 0.1 ms
                                    # for 10M nodes x 7 classes)
REPLAY_RETRIES = 8                  # captures refused because gradients were standing in the buffers (zero_grad(set_to_none=False), a
                                    # second forward before the optimizer step): after this many the inputs stay eager, with one warning


def _stores_of(module):
    """Every parameter store of ``module`` and its sub-modules — found once (walking the F x L sub-modules of a Cora-shaped model
    costs 10 ms; the set of stores is fixed by the constructor)."""
    found = module.__dict__.get("_replay_stores")
    if found is None:
        for m in module.modules():
            if hasattr(m, "_ensure_stores") and "_stores" in m.__dict__:
                m._ensure_stores()
        found = [st for m in module.modules() for st in getattr(m, "_stores", {}).values()]
        object.__setattr__(module, "_replay_stores", found)
    return found


class _Plan:
    def __init__(self, module, inputs, grad: bool):
        from .graphed import GraphedCallable
        self.grad = grad
        self.gen = 0
        dev = inputs.x.device
        stores = _stores_of(module)
        if grad and any(st._occupied(name) for st in stores for name in st.buf):
            raise _NotNow()                             # gradients waiting to be added to: the capture would freeze "assign"
        self.guard = torch.zeros(1, dtype=torch.float32, device=dev)
        # The gradient state as the caller left it — flat gradients (None after the zero_grad of an optimizer over
        # model.parameters(); still standing after one over the per-layer Parameters) and whether the per-layer views are linked
        # — is put back when the plan is built.  In between the flat gradients are None: the captured backward pass must
        # ASSIGN the gradient buffers (``FlatMLPStore._on_grad`` decides "assign or add" in Python, i.e. at capture time);
        # adding to what stands there is the replaying node's business (``_Replayed.backward``).
        before = {(id(st), name): (st.flat[name].grad, st._linked(name)) for st in stores for name in st.buf}

        def blank():
            for st in stores:
                for name in st.buf:
                    st.flat[name].grad = None
                st.pending.clear()
        def restore():
            for st in stores:
                for name in st.buf:
                    was_grad, was_linked = before[(id(st), name)]
                    st.flat[name].grad = was_grad
                    if not was_linked and st._linked(name):
                        for param, _ in st.grad_views.get(name, ()):
                            param.grad = None
                st.pending.clear()
        blank()
        try:
            self._capture(module, inputs, grad, stores, blank)
        finally:
            restore()
        self.ptrs = [st.flat[name].data_ptr() for st in stores for name in st.buf]
        self.anchor = next((st.flat[name] for st in stores for name in st.buf if st.flat[name].requires_grad), None)

    def _capture(self, module, inputs, grad, stores, blank):
        from .graphed import GraphedCallable
        if grad:
            # one more eager forward + backward right before the capture: whatever the backward pass builds on first use (the
            # transposed adjacency, its plans, workspaces) must exist before a capture — which executes nothing — meets it.
            # Its gradients (of a zero upstream gradient) go nowhere.
            with torch.autograd.set_detect_anomaly(False), torch.enable_grad():
                warm = module._forward(inputs)
                torch.autograd.backward(warm, torch.zeros_like(warm))
            del warm
            blank()
        with torch.autograd.set_detect_anomaly(False), torch.set_grad_enabled(grad):
            self.fwd = GraphedCallable(lambda: module._forward(inputs), warmup=0, guard=self.guard)
            self.out = self.fwd.out
            self.guarded = bool(self.fwd.builds)
            self.bwd = self.d_out = None
            self.produced = []
            if grad:
                if not (torch.is_tensor(self.out) and self.out.requires_grad):
                    raise _lib.GnanHipError("the captured forward's output does not require grad")
                self.d_out = torch.zeros_like(self.out)
                for st in stores:
                    st.touched.clear()
                self.bwd = GraphedCallable(lambda: torch.autograd.backward(self.out, self.d_out), warmup=0)
                for st in stores:                       # what the captured backward pass writes (nothing has RUN yet)
                    self.produced += [(st, name) for name in st.buf if name in st.touched]
                    st.touched.clear()

    def stale(self, module) -> bool:
        return [st.flat[name].data_ptr() for st in _stores_of(module) for name in st.buf] != self.ptrs

    def release(self) -> None:
        for g in (self.fwd, self.bwd):
            if g is not None:
                try:
                    g.graph.reset()
                except Exception:
                    pass
                g.out = g.pins = None
        self.out = self.d_out = None


class _NotNow(Exception):
    pass


def _refused(rec) -> None:
    """A capture that found gradients standing in the flat buffers (it would freeze "assign" where "add" is meant).  A loop
    that leaves them there on every forward — ``zero_grad(set_to_none=False)`` — would pay a failed attempt per step for ever:
    after ``REPLAY_RETRIES`` the record is marked dead, once, loudly."""
    rec["refused"] = rec.get("refused", 0) + 1
    if rec["refused"] >= REPLAY_RETRIES and not rec["dead"]:
        rec["dead"] = True
        import warnings
        warnings.warn("gnan_amd.replay: gradients were standing in the parameter buffers at every capture attempt (does the loop call "
                      "zero_grad(set_to_none=False)?): these forwards stay eager.  zero_grad() / zero_grad(set_to_none=True) lets the "
                      "forward and backward replay from hipGraphs")


def _handed_out(plan: "_Plan") -> torch.Tensor:
    out = plan.out.detach()
    if REPLAY_COPY_MAX_BYTES is None or out.numel() * out.element_size() <= REPLAY_COPY_MAX_BYTES:
        return out.clone()
    return out


class _Replayed(torch.autograd.Function):
    """The one autograd node of a replayed forward: its backward replays the captured backward pass."""

    @staticmethod
    def forward(ctx, plan: _Plan, anchor):
        ctx.plan, ctx.gen = plan, plan.gen
        return _handed_out(plan)

    @staticmethod
    def backward(ctx, d_out):
        plan = ctx.plan
        if plan.gen != ctx.gen or plan.bwd is None:
            raise RuntimeError("gnan_amd.replay: this output's forward was replayed again (or released) before its backward "
                               "pass — the activations it reads are gone; call backward() before the next forward() on the same "
                               "inputs, or set gnan_amd.replay.REPLAY_FORWARD = False")
        plan.d_out.copy_(d_out)
        kept = {}
        for st, name in plan.produced:                  # gradients already there (a second backward before zero_grad): added to
            if st._occupied(name):
                kept[(id(st), name)] = st.grad[name].clone()
        plan.bwd.replay()
        for st, name in plan.produced:
            g = st.grad[name]
            if (id(st), name) in kept:
                g.add_(kept[(id(st), name)])
            if not st._linked(name):
                st._link_grads(name)
            if st.flat[name].grad is not g:
                st.flat[name].grad = g
        return None, None


# ---------------------------------------------------------------------------------------------
# the reference's real input cadence: NEW tensor objects every step
# ---------------------------------------------------------------------------------------------
# trainer.py:46 does ``data.to(device)`` on the loader's batch every step: a node task's one graph arrives as fresh device
# tensors each time — same contents, another identity — and an identity-keyed plan never sees its inputs again.  After
# ``REPLAY_AFTER`` such misses with the same shapes the module ADOPTS the inputs: it keeps private copies, captures its plan
# over those (through the ordinary identity route), and every later call with tensors of these shapes is compared with the
# copies on the device (``torch.equal``: one pass over the inputs, one flag); equal contents replay the plan — the output is
# the function of the same numbers — and anything else runs eagerly on the caller's tensors.  Nothing is ever written into the
# adopted copies, so every cache keyed on them (hop-coded graph, value ranges, padded copies) stays valid.
ADOPT_FRESH_INPUTS = True
ADOPT_MAX_BYTES = 1 << 30           # inputs beyond this stay eager (a copy and a comparison per step would cost what the forward costs)
ADOPT_CHANGES_ALLOWED = 8           # contents that keep changing: stop comparing (every call is then eager, as before)
_GRAPH_FIELDS = ("node_distances", "normalization_matrix", "gnan_rowptr", "gnan_col", "gnan_code", "gnan_cnt")


class _AdoptedBook:
    def __init__(self):
        self.entries = {}

    def __deepcopy__(self, memo):
        return _AdoptedBook()

    def __reduce__(self):
        return (_AdoptedBook, ())


class _AdoptedInputs:
    """Duck-typed ``Data`` over the module's private copies of a caller's inputs."""

    def __init__(self, inputs, src):
        self.x = src[0].detach().clone()
        self.edge_index = None
        for name, t in zip(_GRAPH_FIELDS, src[1:]):
            if t is not None:
                setattr(self, name, t.detach().clone())
        for name in ("gnan_graph", "gnan_n_codes"):
            if hasattr(inputs, name):
                setattr(self, name, getattr(inputs, name))

    def tensors(self):
        return (self.x,) + tuple(getattr(self, f, None) for f in _GRAPH_FIELDS)


_MISS = object()


def _run_adopted(module, inputs, src, extra):
    """See ADOPT_FRESH_INPUTS.  Returns the output, or ``_MISS``: the caller goes on with the identity route / the eager forward."""
    shape_key = (tuple(None if t is None else (tuple(t.shape), t.dtype, t.device) for t in src), extra)
    book = module.__dict__.get("_replay_adopted")
    if book is None:
        book = _AdoptedBook()
        object.__setattr__(module, "_replay_adopted", book)
    entry = book.entries.get(shape_key)
    if entry is None:
        if len(book.entries) >= REPLAY_PLANS:
            return _MISS
        book.entries[shape_key] = {"misses": 1, "static": None, "changes": 0}
        return _MISS
    if entry["changes"] > ADOPT_CHANGES_ALLOWED:
        return _MISS
    static = entry["static"]
    if static is None:
        entry["misses"] += 1
        if entry["misses"] <= REPLAY_AFTER or sum(t.numel() * t.element_size() for t in src if t is not None) > ADOPT_MAX_BYTES:
            return _MISS
        static = entry["static"] = _AdoptedInputs(inputs, src)
        return run(module, static)
    same = all(a is None or a is b or torch.equal(a, b) for a, b in zip(src, static.tensors()))
    if not same:
        entry["changes"] += 1
        return _MISS
    return run(module, static)


def _release_record(rec) -> None:
    plan, rec["plan"] = rec.get("plan"), None
    if plan is not None:
        plan.release()


def _key(module, inputs, grad: bool):
    """(source tensors, hashable extras) this forward depends on, or None if it cannot be keyed."""
    x = inputs.x
    g = getattr(inputs, "gnan_graph", None)
    fields = ("node_distances", "normalization_matrix", "gnan_rowptr", "gnan_col", "gnan_code", "gnan_cnt")
    src = (x,) + tuple(t if torch.is_tensor(t) else None for t in (getattr(inputs, f, None) for f in fields))
    flags = tuple(bool(st.flat[name].requires_grad) for st in _stores_of(module) for name in st.buf)
    extra = (grad, bool(module.training), None if g is None else id(g), getattr(module, "aggregation_order", None),
             str(getattr(module, "operand_dtype", None)), bool(getattr(module, "normalize_rho", True)), flags,
             int(getattr(inputs, "gnan_n_codes", 0) or 0))
    return src, extra


class _SlotBook:
    """The slot plans of ONE model (``model.__dict__['_replay_slots']``).  A copied or pickled model starts without plans: a
    hipGraph belongs to the tensors it was captured over (``copy.deepcopy(model)`` is what keeps a best-so-far checkpoint)."""

    def __init__(self):
        self.calls = 0
        self.plans = {}

    def __deepcopy__(self, memo):
        return _SlotBook()

    def __reduce__(self):
        return (_SlotBook, ())


class _SlotInputs:
    """Duck-typed ``Data`` over the slots of a :class:`small_graph.SlotGraph`."""

    def __init__(self, slot):
        self.x, self.edge_index, self.gnan_graph = slot.x, None, slot


def _run_small(module, inputs):
    """A graph-level forward of a SMALL dense graph (batch_size = 1: another graph object every step) through graph slots: the
    graph is copied into the slots of its (node, hop-code) tier — one launch — and the tier's plan, captured over the slots, is
    replayed: whatever graph comes, forward and backward are one graph launch each.  None: not for this route."""
    from .small_graph import SlotGraph, slot_mode, slot_tier
    use_cnt = slot_mode(module)
    if use_cnt is None or inputs.x.dtype != torch.float32:
        return None
    g = module.hop_graph(inputs)
    tier = slot_tier(g)
    if tier is None or g.n_rows != inputs.x.shape[0]:
        return None
    stores = _stores_of(module)
    grad = torch.is_grad_enabled() and any(st.flat[name].requires_grad for st in stores for name in st.buf)
    book = module.__dict__.get("_replay_slots")
    if book is None:
        book = _SlotBook()
        object.__setattr__(module, "_replay_slots", book)
    key = (tier, grad, bool(module.training), tuple(bool(st.flat[name].requires_grad) for st in stores for name in st.buf))
    rec = book.plans.get(key)
    if rec is None:
        if book.calls < REPLAY_AFTER:
            book.calls += 1
            return None
        rec = book.plans[key] = {"plan": None, "slot": None, "dead": False}
    if rec["dead"]:
        return None
    plan = rec["plan"]
    if plan is not None and plan.stale(module):
        plan.release()
        plan = rec["plan"] = None
    if plan is None:
        from .graphed import CaptureFailed
        try:
            slot = SlotGraph(int(inputs.x.shape[1]), inputs.x.device, use_cnt=use_cnt, max_nodes=tier[0], n_codes=tier[1])
            if not slot.fits(g, inputs.x):
                return None
            slot.load(g, inputs.x)
            plan = _Plan(module, _SlotInputs(slot), grad)
            rec["plan"], rec["slot"] = plan, slot
        except _NotNow:
            _refused(rec)
            return None
        except (CaptureFailed, _lib.GnanHipError, RuntimeError) as e:
            rec["dead"] = True
            import warnings
            warnings.warn(f"gnan_amd.replay: small graphs of tier {tier} stay eager ({type(e).__name__}: {str(e)[:200]})")
            return None
    if not rec["slot"].fits(g, inputs.x):              # (another feature count, a graph without the shell sizes the slot holds)
        return None
    rec["slot"].load(g, inputs.x)
    plan.gen += 1
    plan.fwd.replay()
    if plan.guarded and bool(plan.guard.item()):       # (a slot forward that builds tables: as in run())
        plan.guard.zero_()
        _release_record(rec)
        return None
    return _Replayed.apply(plan, plan.anchor) if plan.grad else _handed_out(plan)


def run(module, inputs, node_ids=None):
    """``module.forward``: a replay where a plan exists (or can be captured now), the eager forward otherwise."""
    x = getattr(inputs, "x", None)
    if (not REPLAY_FORWARD or node_ids is not None or not torch.is_tensor(x) or not x.is_cuda or x.requires_grad
            or x.dim() != 2 or getattr(module, "stage_hook", None) is not None
            or module._dropout_active() or torch.cuda.is_current_stream_capturing()):
        return module._forward(inputs) if node_ids is None else module._forward(inputs, node_ids)
    if x.shape[0] < REPLAY_MIN_NODES:
        out = None
        if REPLAY_SMALL_GRAPHS and getattr(inputs, "gnan_graph", None) is None:
            out = _run_small(module, inputs)
        return module._forward(inputs) if out is None else out
    grad = torch.is_grad_enabled() and any(st.flat[name].requires_grad for st in _stores_of(module) for name in st.buf)
    cache = module.__dict__.get("_replays")
    if cache is None:
        from ._cache import TensorKeyedCache
        cache = TensorKeyedCache(REPLAY_PLANS, on_evict=_release_record)
        object.__setattr__(module, "_replays", cache)
    src, extra = _key(module, inputs, grad)
    rec = cache.get(src, extra)
    if rec is None and ADOPT_FRESH_INPUTS and not isinstance(inputs, _AdoptedInputs):
        out = _run_adopted(module, inputs, src, extra)
        if out is not _MISS:
            return out
    if rec is None:
        if len(cache) >= cache.capacity:
            # room only at the expense of an entry that holds no plan yet (the least recently used such one)
            spare = next((k for k, e in cache.entries.items() if e.value["plan"] is None), None)
            if spare is None:
                return module._forward(inputs)
            del cache.entries[spare]
        cache.put(src, extra, {"calls": 1, "plan": None, "dead": False, "graph": getattr(inputs, "gnan_graph", None)})
        return module._forward(inputs)
    if rec["dead"]:
        return module._forward(inputs)
    plan: Optional[_Plan] = rec["plan"]
    if plan is not None and plan.stale(module):
        _release_record(rec)
        plan, rec["calls"] = None, REPLAY_AFTER
    if plan is None:
        if rec["calls"] < REPLAY_AFTER:
            rec["calls"] += 1
            return module._forward(inputs)
        from .graphed import CaptureFailed
        try:
            plan = rec["plan"] = _Plan(module, inputs, grad)
        except _NotNow:
            _refused(rec)
            return module._forward(inputs)
        except (CaptureFailed, _lib.GnanHipError, RuntimeError) as e:
            rec["dead"] = True
            import warnings
            warnings.warn(f"gnan_amd.replay: this forward stays eager ({type(e).__name__}: {str(e)[:200]})")
            return module._forward(inputs)
    plan.gen += 1
    plan.fwd.replay()
    if plan.guarded and bool(plan.guard.item()):        # the tables of these weights outgrew the captured look-up: its output is void
        plan.guard.zero_()
        _release_record(rec)
        rec["calls"] = REPLAY_AFTER                     # (captured again next time, with the sizes this eager forward leaves)
        return module._forward(inputs)
    if plan.grad:
        return _Replayed.apply(plan, plan.anchor)
    return _handed_out(plan)


def release(module) -> None:
    """Drop every plan of ``module`` (their hipGraphs and private memory pools)."""
    cache = module.__dict__.pop("_replays", None)
    if cache is not None:
        for e in list(cache.entries.values()):
            _release_record(e.value)
        cache.clear()
    book = module.__dict__.pop("_replay_slots", None)
    if book is not None:
        for rec in book.plans.values():
            _release_record(rec)
    module.__dict__.pop("_replay_adopted", None)
