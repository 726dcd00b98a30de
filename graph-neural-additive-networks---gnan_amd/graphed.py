"""hipGraph replay of whole forward / training steps on a FIXED input (node-level tasks: one graph, every epoch).

The reference's node-level loop (trainer.py:23-86 on Cora / ogbn-arxiv) runs the same computation on the same tensors
every epoch: zero the gradients, forward, mask, loss, backward, Adam.  On an MI355X the kernels of such a step take
~1 ms on the arxiv-shaped graph while Python, the autograd engine and ~70 kernel launches take 2.5 ms — the step is
bound by the host.  A hipGraph removes the host from the loop: the step is captured ONCE (all launches of this library
go to the capturing stream like any other; the table build's piece counts, the one value a forward normally reads
back, stay on the device) and every further epoch is a single ``graph.replay()``.

What a captured step freezes, and how it is guarded:
* tensor ADDRESSES of the inputs, parameters and optimizer state — :meth:`GraphedStep.stale` compares the parameters'
  addresses and the hyper-parameters before every replay; ``harness`` re-captures when it reports a change;
* the SIZES of the shape-function tables (search depth, LDS image) — they follow the weights, which training moves;
  the captured look-up has the same head-room as the speculative look-up of the eager path and checks, on the device, that
  the tables its own build produced fit it (``gnan_pwl_check_fit`` sets the step's guard flag); the captured optimizer
  update takes the flag as its skip flag (torch's ``found_inf``), so a step whose tables outgrew the capture changes
  nothing; the flag is read after the replay and such a step is run eagerly and captured again.  (Optimizers whose
  update is not the flat fused one cannot be skipped on the device: :meth:`GraphedCallable.fits` then builds the tables
  of the current weights BEFORE every replay — one more table build, a read-back and a host synchronisation per epoch;)
* the learning rate lives in a device tensor the schedulers' floats are copied into before each replay.
Dropout in training mode is stochastic per call and is never captured (``GraphedStep.supported``).
"""
from __future__ import annotations

import atexit
import weakref
from typing import Callable, List, Optional

import torch

from . import _lib, functional


class CaptureFailed(RuntimeError):
    pass


# Captured graphs hold kernel nodes that point into libgnan_hip.so (the memset replacements of csrc/graph_fix.hip) and live in
# the framework's private pools.  At interpreter shutdown the order in which the library, the HIP runtime and the graph objects
# go away is not ours to choose (one unexplained crash at exit in ~25 test runs); they are released here first, while
# everything they refer to is still loaded.
_LIVE = weakref.WeakSet()


def _release_graphs_at_exit():
    for g in list(_LIVE):
        try:
            g.graph.reset()
        except Exception:
            pass
    try:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass


atexit.register(_release_graphs_at_exit)


class GraphedCallable:
    """``fn()`` — no arguments, closes over static tensors — captured into a hipGraph after ``warmup`` eager runs."""

    def __init__(self, fn: Callable[[], object], warmup: int = 2, before_capture: Optional[Callable[[], None]] = None,
                 guard: Optional[torch.Tensor] = None):
        dev = torch.cuda.current_device()
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):              # lazy initialisations, caches (hop-graph plans, table sizes), allocator
                fn()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        if before_capture is not None:
            before_capture()
        functional.CAPTURED_BUILDS.clear()
        # tensors the step reads that are owned by nobody but an LRU cache (max |x| of the fixed-point scales, the padded
        # feature matrix, the hop-coded graph derived from the inputs): the graph bakes their ADDRESSES in, so the step
        # keeps them alive — an eviction would hand their blocks to someone else under a replay
        functional.CAPTURE_PINS = pins = []
        functional.CAPTURE_GUARD = guard
        functional.CAPTURE_SCRATCH = self.scratch = torch.zeros(1 << 16, dtype=torch.int32, device=dev)   # eager, owned by this step
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        # with a process group up, its watchdog thread polls events of earlier collectives while this thread captures: a
        # "global" capture would be invalidated by those calls, a thread-local one is not
        import torch.distributed as dist
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        try:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.out = fn()
            # a captured hipMemsetAsync replays correctly only once on this ROCm (csrc/graph_fix.hip); the framework's
            # multi-block reductions (a loss's mean, a column sum) zero their semaphores with one: swap the nodes for kernels
            import ctypes
            swapped = ctypes.c_int32(0)
            _lib.check(_lib.lib().gnan_graph_replace_memsets(self.graph.raw_cuda_graph(), ctypes.byref(swapped)),
                       "gnan_graph_replace_memsets")
            self.memsets_replaced = int(swapped.value)
            kernels = ctypes.c_int32(0)
            _lib.check(_lib.lib().gnan_graph_node_count(self.graph.raw_cuda_graph(), ctypes.byref(kernels), None),
                       "gnan_graph_node_count")
            self.kernel_nodes = int(kernels.value)          # launches per replay
            self.graph.instantiate()
        except Exception as e:                   # a host synchronisation on the path, an unsupported op, ...
            functional.CAPTURED_BUILDS.clear()
            functional.CAPTURE_PINS = functional.CAPTURE_GUARD = functional.CAPTURE_SCRATCH = None
            raise CaptureFailed(f"{type(e).__name__}: {e}") from e
        self.builds = list(functional.CAPTURED_BUILDS)
        functional.CAPTURED_BUILDS.clear()
        functional.CAPTURE_PINS = functional.CAPTURE_GUARD = functional.CAPTURE_SCRATCH = None
        self.pins = pins
        self.replays = 0
        _LIVE.add(self)

    def replay(self):
        self.graph.replay()
        self.replays += 1
        return self.out

    def fits(self) -> bool:
        """Do the shape-function tables of the CURRENT weights fit the sizes the look-ups were captured with?"""
        from . import pwl
        for spec, stacked in self.builds:
            tables = pwl.build_tables(stacked)
            if tables is None or not pwl.covers(spec, tables):
                return False
        return True


def _probe_params(model: torch.nn.Module):
    """A few Parameter objects whose storage addresses stand for the whole module's (``.to()`` and ``FlatMLPStore.rebuild``
    move all of them together); walking all F x L parameters before every replay would cost more than the replay."""
    ps = list(model.parameters())
    return [ps[i] for i in sorted({0, len(ps) // 3, len(ps) // 2, (2 * len(ps)) // 3, len(ps) - 1})] if ps else []


def _group_signature(optimizer):
    return tuple(tuple(sorted((k, v) for k, v in g.items() if k not in ("params", "lr") and isinstance(v, (int, float, bool, tuple, type(None)))))
                 for g in optimizer.param_groups)


class PreparedOptimizer:
    """An Adam-family optimizer switched to the mode a captured step needs — step counters and learning rate on the device
    (``capturable=True``), the fused multi-tensor update — together with what it was before.

    The switch changes the caller's object: ``group['lr']`` becomes a device tensor, ``optimizer.state_dict()`` carries it
    and device ``step`` counters, and the update arithmetic is the fused kernel's.  It is therefore made ONCE, before the
    first epoch of a loop that will be captured (``harness`` does so on its first call), never in the middle of a run, and
    :meth:`restore` gives the optimizer its own flags and a float learning rate back — when a capture fails, when the
    captured steps are released, or on request (e.g. before writing a checkpoint the reference's trainer should read)."""

    def __init__(self, optimizer):
        # a STRONG reference (the optimizer does not own the model, so this keeps no model alive): the cycle collector clears
        # weak references held by the garbage it is about to finalise, and restore() has to work from such a finaliser
        self.optimizer = lambda: optimizer
        self.saved = []
        self.lrs: List[torch.Tensor] = []
        for group in optimizer.param_groups:
            if "capturable" not in group:
                raise CaptureFailed(f"{type(optimizer).__name__} has no capturable mode")
        for group in optimizer.param_groups:
            self.saved.append({k: group.get(k) for k in ("capturable", "fused", "foreach", "lr") if k in group})
            group["capturable"] = True
            if "fused" in group and all(p.is_cuda and torch.is_floating_point(p) for p in group["params"]):
                # F x L small parameter tensors (774 on the arxiv shape): the for-each implementation needs ~15 launches per
                # 30 tensors and falls back to one launch per tensor for the operations that take the learning-rate TENSOR
                # (1300 launches, 4 ms per replayed step); the fused kernel updates 36 tensors per launch in one pass
                group["fused"], group["foreach"] = True, False
            dev = group["params"][0].device
            lr = group["lr"]
            static = lr if torch.is_tensor(lr) and lr.device == dev else torch.tensor(float(lr), dtype=torch.float32, device=dev)
            group["lr"] = static
            self.lrs.append(static)
            for p in group["params"]:
                st = optimizer.state.get(p)
                if st and "step" in st and torch.is_tensor(st["step"]) and st["step"].device != p.device:
                    st["step"] = st["step"].to(p.device, torch.float32)

    def restore(self) -> None:
        opt = self.optimizer()
        if opt is None or self.saved is None:
            return
        for group, saved, static in zip(opt.param_groups, self.saved, self.lrs):
            now = group["lr"]
            group.update(saved)                               # incl. the caller's own learning-rate object ...
            if torch.is_tensor(now):                          # ... unless a scheduler has moved the rate since
                was = saved.get("lr")
                same = (not torch.is_tensor(was)) and float(torch.tensor(float(was), dtype=torch.float32)) == float(now)
                if not same:
                    group["lr"] = float(now)
            else:
                group["lr"] = now
            if not saved.get("capturable", False):
                for p in group["params"]:                     # the non-capturable update wants its step counters on the host
                    st = opt.state.get(p)
                    if st and torch.is_tensor(st.get("step")) and st["step"].is_cuda:
                        st["step"] = st["step"].detach().cpu()
        self.saved = None


def prepare_optimizer(optimizer) -> PreparedOptimizer:
    return PreparedOptimizer(optimizer)


GUARDED_REPLAY = True        # table-size check inside the captured step instead of before every replay
FLAT_OPTIMIZER_STEP = True   # captured steps update the FlatMLPStore buffers with one fused launch


class FlatAdamStep:
    """``optimizer.step()`` of a fused Adam / AdamW over a model whose Parameters are views of ``FlatMLPStore`` buffers, as
    ONE fused update over the flat buffers.

    The reference model owns F x L tiny ``nn.Linear`` tensors (8604 on the Cora shape, 1170 on the arxiv shape) and torch's
    fused update takes ~70 of them per launch: 120 launches = 0.86 ms of a 4.9-ms replayed Cora-shaped step, 30 launches =
    0.18 of 0.9 ms on the arxiv shape.  The Parameters and their gradients already lie in a dozen contiguous buffers; here
    the optimizer STATE is re-homed the same way — ``state[p]['exp_avg']``, ``['exp_avg_sq']`` and ``['step']`` become
    views of flat tensors, values kept — and the update is one ``torch._fused_adam_`` call over the flat tensors: the same
    kernel applied to the same elements, bit for bit what ``optimizer.step()`` computes, and ``optimizer.step()`` itself,
    ``state_dict()`` and ``load_state_dict()`` keep working on the views.

    :meth:`build` returns None whenever the equivalence is not obvious: another optimizer class, several parameter groups,
    ``amsgrad`` / ``maximize`` / ``differentiable``, parameters outside the stores, gradients missing, step counters that
    differ between parameters."""

    def __init__(self):
        self.P, self.G, self.M, self.V, self.step_of = [], [], [], [], []
        self.steps = None
        self.decoupled = False

    @staticmethod
    def build(model, optimizer) -> Optional["FlatAdamStep"]:
        if not FLAT_OPTIMIZER_STEP or type(optimizer) not in (torch.optim.Adam, torch.optim.AdamW):
            return None
        if len(optimizer.param_groups) != 1:
            return None
        group = optimizer.param_groups[0]
        if group.get("amsgrad") or group.get("maximize") or group.get("differentiable") or not group.get("fused"):
            return None
        stores = [st for m in model.modules() for st in getattr(m, "_stores", {}).values()]
        names = [(st, name) for st in stores for name in st.buf]
        if not names or any(not st.consistent() or name not in st.grad for st, name in names):
            return None
        flats = [st.flat[name] for st, name in names]
        if {id(p) for p in group["params"]} == {id(p) for p in flats} and len(group["params"]) == len(flats):
            return FlatAdamStep._over_flat_parameters(optimizer, names, flats)
        pairs = [st.views_like(name, st.buf[name]) for st, name in names]
        covered = {id(p) for pr in pairs for p, _ in pr}
        if covered != {id(p) for p in group["params"]} or len(covered) != sum(len(pr) for pr in pairs):
            return None
        if any(p.grad is not None and p.grad.is_sparse for p in group["params"]):
            return None
        self = FlatAdamStep()
        self.decoupled = bool(group.get("decoupled_weight_decay", False))
        dev = names[0][0].buf[names[0][1]].device
        state = optimizer.state
        seen = [state[p]["step"] for p in group["params"] if p in state and "step" in state[p]]
        if seen and len(seen) != len(covered):
            return None                                  # some parameters have stepped and some have not
        if seen:
            all_steps = torch.stack([s.detach().to(dev, torch.float32).reshape(()) for s in seen])
            if bool((all_steps != all_steps[0]).any()):
                return None
            start = all_steps[0].clone()
        else:
            start = torch.zeros((), dtype=torch.float32, device=dev)
        self.steps = start.repeat(len(covered)).contiguous()       # one float32 counter per Parameter, as the fused update keeps them
        at = 0
        for (st, name), pr in zip(names, pairs):
            P, G = st.buf[name], st.grad[name]
            M, V = torch.zeros_like(P), torch.zeros_like(P)
            for (p, _), (_, m), (_, v) in zip(pr, st.views_like(name, M), st.views_like(name, V)):
                old = state.get(p)
                if old and "exp_avg" in old:
                    m.copy_(old["exp_avg"])
                    v.copy_(old["exp_avg_sq"])
                state[p] = {"step": self.steps[at], "exp_avg": m, "exp_avg_sq": v}
                at += 1
            self.P.append(P); self.G.append(G); self.M.append(M); self.V.append(V)
            self.step_of.append(self.steps[at - 1])
        self.optimizer = optimizer
        self.whole = list(zip(self.P, self.G))                       # the stores' own tensors (valid() compares identities)
        self._split()
        return self

    @staticmethod
    def _over_flat_parameters(optimizer, names, flats) -> Optional["FlatAdamStep"]:
        """The optimizer was built over ``model.parameters()`` — the flat Parameters themselves (``modules.FLAT_PARAMETERS``):
        its own state tensors ARE flat already.  What this object adds to ``optimizer.step()`` is the skip flag of a guarded
        replay and the cut into chunk-sized pieces (:meth:`_split`); state that does not exist yet is created as the optimizer
        would create it (zeros, step 0)."""
        group = optimizer.param_groups[0]
        state = optimizer.state
        if any(p.grad is not None and p.grad.is_sparse for p in flats):
            return None
        self = FlatAdamStep()
        self.decoupled = bool(group.get("decoupled_weight_decay", False))
        dev = flats[0].device
        seen = [state[p]["step"] for p in flats if p in state and "step" in state[p]]
        if seen and len(seen) != len(flats):
            return None
        if seen:
            all_steps = torch.stack([s.detach().to(dev, torch.float32).reshape(()) for s in seen])
            if bool((all_steps != all_steps[0]).any()):
                return None
            start = all_steps[0].clone()
        else:
            start = torch.zeros((), dtype=torch.float32, device=dev)
        self.steps = start.repeat(len(flats)).contiguous()
        for at, ((st, name), fp) in enumerate(zip(names, flats)):
            old = state.get(fp)
            if old and "exp_avg" in old:
                M, V = old["exp_avg"], old["exp_avg_sq"]
                if M.shape != fp.shape or not M.is_contiguous() or not V.is_contiguous() or M.device != fp.device:
                    return None
            else:
                M, V = torch.zeros_like(fp.data), torch.zeros_like(fp.data)
            state[fp] = {"step": self.steps[at], "exp_avg": M, "exp_avg_sq": V}
            self.P.append(st.buf[name]); self.G.append(st.grad[name]); self.M.append(M); self.V.append(V)
            self.step_of.append(self.steps[at])
        self.optimizer = optimizer
        self.whole = list(zip(self.P, self.G))
        self._split()
        return self

    def _split(self, target: int = 24, floor: int = 4096) -> None:
        """The multi-tensor kernel gives every 65 536-element chunk of a tensor to ONE workgroup: a small model's flat
        buffers would be updated by a dozen workgroups, each walking tens of thousands of elements (the 15-feature graph-task
        model: +19 us per step against its 96 separate tensors).  Buffers are therefore cut into about ``target`` pieces of
        at least ``floor`` elements in total — still one launch (36 tensors fit), but the pieces of a small model are short;
        large models keep chunk-sized work per workgroup."""
        total = sum(t.numel() for t in self.P)
        piece = max(floor, -(-total // target))
        piece = -(-piece // 4) * 4                                   # whole 16-byte vectors
        lists = ([], [], [], [], [])
        for P, G, M, V, st in zip(self.P, self.G, self.M, self.V, self.step_of):
            flat = [t.view(-1) for t in (P, G, M, V)]
            for lo in range(0, P.numel(), piece):
                for dst, t in zip(lists, flat):
                    dst.append(t[lo:lo + piece])
                lists[4].append(st)
        self.buffers = len(self.P)
        self.P, self.G, self.M, self.V, self.step_of = lists

    def valid(self, model, optimizer) -> bool:
        """Still the flat form of THIS optimizer over THIS model's current buffers?  (Every captured step of a model — one
        per graph shape in a graph-level task — shares one instance: a second ``build`` would re-home the state again and
        strand the steps captured over the first.)"""
        if optimizer is not self.optimizer or not self.intact(optimizer):
            return False
        stores = [st for m in model.modules() for st in getattr(m, "_stores", {}).values()]
        now = [(st.buf[name], st.grad.get(name)) for st in stores for name in st.buf]
        return len(now) == len(self.whole) and all(a is p and b is g for (a, b), (p, g) in zip(now, self.whole))

    def intact(self, optimizer) -> bool:
        """The optimizer state still consists of the flat tensors' views (``load_state_dict`` replaces the state tensors)."""
        group = optimizer.param_groups[0]
        p0, p1 = group["params"][0], group["params"][-1]
        s0, s1 = optimizer.state.get(p0), optimizer.state.get(p1)
        lo, hi = self.steps.data_ptr(), self.steps.data_ptr() + self.steps.numel() * 4
        return bool(s0 and s1 and lo <= s0["step"].data_ptr() < hi and lo <= s1["step"].data_ptr() < hi)

    def step(self, skip: Optional[torch.Tensor] = None) -> None:
        """``skip``: float32 [1] device flag — 1.0 makes the kernel leave parameters and moments alone (torch's ``found_inf``);
        the step counters are advanced regardless (:meth:`undo_count` takes that back)."""
        g = self.optimizer.param_groups[0]
        self.steps.add_(1)
        fn = torch._fused_adamw_ if self.decoupled else torch._fused_adam_
        beta1, beta2 = g["betas"]
        fn(self.P, self.G, self.M, self.V, [], self.step_of, amsgrad=False, lr=g["lr"], beta1=float(beta1), beta2=float(beta2),
           weight_decay=g["weight_decay"], eps=g["eps"], maximize=False, grad_scale=None, found_inf=skip)

    def undo_count(self) -> None:
        self.steps.sub_(1)


class GraphedStep:
    """One captured step on fixed inputs: ``outputs = model(data)``, optionally followed by
    ``loss = loss_of(outputs)``, ``loss.backward()``, ``optimizer.step()``.

    ``loss_of(outputs) -> (loss, extras)`` must be free of host synchronisations (index with precomputed integer
    indices, not boolean masks).  After :meth:`replay`, ``self.outputs``, ``self.loss`` and ``self.extras`` hold the
    replay's results in static tensors."""

    @staticmethod
    def supported(model, optimizer=None) -> bool:
        if not torch.cuda.is_available():
            return False
        if getattr(model, "_dropout_active", lambda: False)():
            return False
        return optimizer is None or all("capturable" in g for g in optimizer.param_groups)

    def __init__(self, model, data, loss_of=None, optimizer=None, forward: Optional[Callable] = None, warmup: int = 0,
                 prepared: Optional[PreparedOptimizer] = None, guard: Optional[torch.Tensor] = None):
        """``warmup`` eager steps are run first — REAL steps (they update the parameters when an optimizer is given).  A
        capture needs the step to have run eagerly at least once (lazy initialisations, table sizes); ``harness`` passes
        0 because its first epochs already did.  ``prepared``: the optimizer's capturable mode if the caller switched it on
        already (:func:`prepare_optimizer`); otherwise it is switched on here and undone if the capture fails."""
        self.model, self.data, self.optimizer = model, data, optimizer
        self.training = optimizer is not None
        fwd = forward or (lambda: model.forward(data))
        self.prepared = prepared if self.training else None
        self._own_prepared = False
        if self.training and self.prepared is None:
            self.prepared, self._own_prepared = PreparedOptimizer(optimizer), True
        self.lrs = self.prepared.lrs if self.training else []
        self.outputs = self.loss = self.extras = None
        self.flat = None

        # Guarded steps: instead of building the tables of the current weights BEFORE every replay (one more table build, a
        # read-back and a host synchronisation: 0.14 ms of a 0.7-ms arxiv-shaped epoch, 0.5 of 2.1 ms on the Cora shape) the
        # captured look-up checks its own tables on the device (gnan_pwl_check_fit), the captured update is skipped by the
        # kernel when they outgrew the capture (torch's found_inf), and the flag is read AFTER the replay — next to the loss the
        # caller reads anyway.  Needs the flat update (the optimizer's own step takes no skip flag); evaluation steps always can.
        # (``guard``: the caller's own zeroed float32 [1] — it then reads the flag itself, next to its other results, and
        # tells :meth:`replay` through ``tripped``)
        self.guard = (guard if guard is not None else
                      torch.zeros(1, dtype=torch.float32, device=next(model.parameters()).device)) if GUARDED_REPLAY else None

        # (the guard is never reset: a step whose guard tripped is dropped by its replayer, and a fresh capture gets a fresh flag)
        one = torch.ones((), dtype=torch.float32, device=next(model.parameters()).device) if self.training else None

        def step():
            nonlocal one
            if self.training:
                out = fwd()
                out = out[0] if isinstance(out, tuple) else out
                loss, extras = loss_of(out)
                if loss.dim() == 0 and loss.dtype == torch.float32:
                    if one is None or one.device != loss.device:
                        one = torch.ones((), dtype=torch.float32, device=loss.device)
                    torch.autograd.backward(loss, one)       # (loss.backward() fills a fresh tensor of ones every step: a launch)
                else:
                    loss.backward()
                self.flat.step(skip=self.guard) if self.flat is not None else optimizer.step()
                return out.detach(), loss.detach(), extras
            with torch.no_grad():
                out = fwd()
                out = out[0] if isinstance(out, tuple) else out
                loss, extras = loss_of(out) if loss_of is not None else (None, None)
            return out, loss, extras

        def eager_step():
            if self.training:
                optimizer.zero_grad(set_to_none=True)
            return step()

        # gradients must be None when the capture starts: the captured backward then ASSIGNS them (rewritten by every
        # replay) instead of accumulating into the previous epoch's
        clear = (lambda: optimizer.zero_grad(set_to_none=True)) if self.training else None
        self.eager_step = eager_step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self.warmup_result = None                    # (outputs, loss, extras) of the LAST warm-up step: it was a real step
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.warmup_result = eager_step()
        torch.cuda.current_stream().wait_stream(side)
        # the update as one launch over the flat parameter buffers where that is the same computation (FlatAdamStep)
        if self.training:
            flat = getattr(self.prepared, "flat", None)
            if flat is None or not flat.valid(model, optimizer):
                flat = FlatAdamStep.build(model, optimizer)
            self.flat = self.prepared.flat = flat
        if self.training and self.flat is None:
            self.guard = None                     # the optimizer's own update cannot be skipped on the device: check before replaying
        try:
            self.graph = GraphedCallable(step, warmup=0, before_capture=clear, guard=self.guard)
        except CaptureFailed as e:
            if self._own_prepared:
                self.prepared.restore()
            e.warmup_result = self.warmup_result      # warm-up steps were REAL steps: the caller must not step this input again
            raise
        self.outputs, self.loss, self.extras = self.graph.out
        self._probes = _probe_params(model)
        self._params = [p.data_ptr() for p in self._probes]
        self._groups = _group_signature(optimizer) if self.training else None
        self._mode = model.training

    def release(self, restore_optimizer: bool = True) -> None:
        """Free the captured graph (and its private memory pool) and drop what the step kept alive; with
        ``restore_optimizer`` the optimizer gets its own settings back (:meth:`PreparedOptimizer.restore`)."""
        g, self.graph = getattr(self, "graph", None), None
        if g is not None:
            try:
                g.graph.reset()
            except Exception:
                pass
            g.out = g.pins = None
            _LIVE.discard(g)
        if restore_optimizer and self.prepared is not None:
            self.prepared.restore()
        self.outputs = self.loss = self.extras = None
        self.model = self.data = self.optimizer = self.eager_step = self.flat = None

    def stale(self) -> bool:
        """Something the graph froze has changed: parameter storage (``.to()``, ``load_state_dict`` into new tensors),
        optimizer hyper-parameters other than the learning rate, train/eval mode."""
        if [p.data_ptr() for p in self._probes] != self._params or self.model.training != self._mode:
            return True
        if self.training and self.flat is not None and not self.flat.intact(self.optimizer):
            return True
        return self.training and _group_signature(self.optimizer) != self._groups

    def replay(self, tripped: Optional[Callable[[], bool]] = None):
        """Replay the captured step; returns ``(outputs, loss, extras)`` (static tensors), or None if the step must not be
        replayed (see :meth:`stale`, :meth:`GraphedCallable.fits`) — the caller then runs it eagerly and captures anew.
        ``tripped``: called after the replay instead of reading the guard flag here (the caller reads it together with its
        own results: one device-to-host copy instead of two)."""
        if self.graph is None or self.stale():
            return None
        guarded = self.guard is not None and bool(self.graph.builds)
        if not guarded and not self.graph.fits():
            return None
        for group, static in zip(self.optimizer.param_groups if self.training else [], self.lrs):
            if group["lr"] is not static:                     # a scheduler wrote a float: keep the tensor, take the value
                static.fill_(float(group["lr"]))
                group["lr"] = static
        self.graph.replay()
        if guarded and (tripped() if tripped is not None else bool(self.guard.item())):
            # the tables of these weights outgrew the captured look-up: its outputs are not the model's, and the update was
            # skipped on the device; only the step counters moved
            if self.training:
                self.flat.undo_count()
            return None
        return self.outputs, self.loss, self.extras


class _Slots:
    """Duck-typed ``Data`` whose tensors are the static input buffers of a captured graph-task step."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class SlottedGraphStep:
    """A captured training step for ALL graphs of one shape ``(nodes, features, shells)`` of a graph-level task.

    Graph-level tasks see a different small graph every step (trainer.py:23-86 with batch_size = 1), so a captured step
    cannot be tied to one input.  It is tied to a SHAPE instead: the step is captured over static buffers (feature matrix,
    dense hop codes, shell counts, label) and each graph of that shape is copied into them — four tiny device copies — before
    the replay.  ``loss_of(outputs, label) -> loss`` as in :class:`GraphedStep`; the caller's running totals are updated
    inside the captured step (``totals = (loss_sum, hit_count)``, device scalars)."""

    def __init__(self, model, optimizer, loss_of, graph, x, label, prepared: Optional[PreparedOptimizer] = None):
        from .graph import HopGraph
        if not graph.is_dense:
            raise CaptureFailed("graph-task steps are captured for the dense layout only")
        self.x, self.code, self.cnt = torch.empty_like(x), torch.empty_like(graph.code), torch.empty_like(graph.cnt)
        self.label = torch.empty_like(label)
        self.load(graph, x, label)
        static_graph = HopGraph(n_rows=graph.n_rows, n_cols=graph.n_cols, n_codes=graph.n_codes, code=self.code, cnt=self.cnt)
        self.data = _Slots(x=self.x, edge_index=None, gnan_graph=static_graph)
        self.step = GraphedStep(model, self.data, lambda out: loss_of(out, self.label), optimizer, warmup=0, prepared=prepared)

    def load(self, graph, x, label) -> None:
        pairs = [(self.x, x), (self.code, graph.code), (self.cnt, graph.cnt), (self.label, label)]
        if all(s.is_contiguous() and s.dtype == d.dtype and s.shape == d.shape and s.is_cuda for d, s in pairs):
            _lib.multi_copy(pairs)                    # one launch instead of four
            return
        for d, s in pairs:
            d.copy_(s)

    def run(self, graph, x, label):
        """Copy the graph into the slots and replay; None if the capture has gone stale (the caller steps eagerly)."""
        if self.step.graph is None or self.step.stale():
            return None
        self.load(graph, x, label)
        return self.step.replay()


class SlotGraphStep:
    """ONE captured training (or evaluation) step for EVERY graph of a graph-level task that fits the slots of a
    ``small_graph.SlotGraph`` (up to 128 nodes, 63 hops): the step is captured over the slots, whose kernels read the graph's
    size from the device, and each graph is copied into them — one launch — before the replay.  No capture per graph shape:
    the first epoch already replays, and the memory a run reserves does not grow with the number of shapes it meets
    (``SlottedGraphStep``: 451 captures / 2 GB on the Mutagenicity-shaped set)."""

    def __init__(self, model, optimizer, loss_of, n_features: int, label: torch.Tensor, use_cnt: bool,
                 prepared: Optional[PreparedOptimizer] = None, first=None, max_nodes: int = 128, n_codes: int = 64):
        from .small_graph import SlotGraph
        dev = label.device
        self.slot = SlotGraph(n_features, dev, use_cnt=use_cnt, max_nodes=max_nodes, n_codes=n_codes)
        self.label = torch.empty_like(label)
        if first is not None:                         # (graph, x, label): what the capture's eager pass runs on
            self.load(*first)
        self.data = _Slots(x=self.slot.x, edge_index=None, gnan_graph=self.slot)
        self.step = GraphedStep(model, self.data, lambda out: loss_of(out, self.label), optimizer, warmup=0, prepared=prepared)

    def fits(self, graph, x, label) -> bool:
        return self.slot.fits(graph, x) and label.shape == self.label.shape and label.dtype == self.label.dtype

    def load(self, graph, x, label) -> None:
        self.slot.load(graph, x, extra=[(self.label, label)])

    def run(self, graph, x, label):
        """Copy the graph into the slots and replay; None if it does not fit or the capture has gone stale."""
        if self.step.graph is None or self.step.stale() or not self.fits(graph, x, label):
            return None
        self.load(graph, x, label)
        return self.step.replay()
