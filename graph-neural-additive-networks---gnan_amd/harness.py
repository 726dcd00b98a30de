"""Epoch loops with the call contract of the reference's ``trainer.py``.

``train_epoch`` / ``test_epoch`` take the same arguments and return the same tuples as trainer.py:23-86 /
:89-154, so ``main.py``-style drivers can switch by changing one import.  What they reproduce: label handling
(trainer.py:32-40), the node-task mask applied AFTER the forward (:53-55, :125-131), the flatten rule of the
loss (:61-64), accuracy (:5-20) and ROC-AUC (:78,147), the fact that ``test_epoch`` leaves the model in eval
mode (:97).  What they do differently, for the GPU: batches that already live on ``device`` are not copied
again (the reference re-uploads both N x N matrices on every call, trainer.py:46,109), losses and hit counts
are accumulated on the device and read back once per epoch instead of one ``.item()`` sync per batch, and the
autograd anomaly mode that wraps every reference epoch (trainer.py:24) is not switched on.

Node-level tasks run the same step on the same tensors every epoch (one full-batch graph).  When the loader hands out
ONE batch that already lives on the device — the same objects every epoch, e.g. ``[data.to(device)]`` — the third and
every later epoch replays a hipGraph of the whole step (forward, mask, loss, backward, Adam update, hit count;
``gnan_amd/graphed.py``) instead of re-issuing ~70 launches from Python: same arithmetic, same kernels, no host in the
loop.  Graph-level tasks (a different small graph per step, batch_size = 1) get one captured step per graph SHAPE
(nodes, features, shells): the third graph of a shape captures it over static buffers and every later graph of that shape
is copied into them and replayed (``SlottedGraphStep``).  ``GNAN_GRAPHED_STEPS=0`` switches it off; anything a capture cannot hold (training-mode Dropout, an optimizer
without a capturable mode, a loader that uploads fresh tensors per epoch) silently keeps the eager loop.

The loss step itself — rows of the task mask, ``nn.BCEWithLogitsLoss()`` / ``nn.CrossEntropyLoss()`` at their default options,
the gradient w.r.t. the logits, the hit count and the epoch's running totals — is one launch (``losses.loss_step`` ->
``gnan_loss_step``) in the eager loop and in the captured steps alike; any other loss callable runs as given
(``GNAN_FUSED_LOSS=0``: always).
"""
from __future__ import annotations

import os
import warnings
import weakref

import numpy as np
import torch

GRAPHED_STEPS = os.environ.get("GNAN_GRAPHED_STEPS", "1") != "0"
GRAPH_AFTER = 2                      # eager epochs before a step is captured (they are the capture's warm-up)
GRAPH_TASK_MAX_SHAPES = 4096         # captured steps a graph-level task may hold (each owns a private memory pool: >= 2 MB)
SLOT_STEPS = True                    # graph-level tasks: ONE captured step over graph slots (graphed.SlotGraphStep) for every graph that
                                     # fits them (<= 128 nodes, <= 63 hops); per-shape steps only for the others
SLOT_AFTER = 2                       # eager steps of the loop before the slot step is captured (lazy initialisations, table sizes)


class _StepStore:
    """Captured steps of ONE model, kept on the model instance itself (``model.__dict__['_gnan_steps']``): a captured step
    holds its model, data and optimizer strongly, so a registry keyed by the model — even a weak one — would keep every
    model of a cross-validation loop (main.py builds a fresh one per fold and seed) and its graphs' private memory pools
    alive.  Hung off the model, records and model form a cycle the garbage collector frees when the model is dropped.
    A copied or pickled model starts without captured steps."""

    def __init__(self):
        from ._cache import TensorKeyedCache
        # per-(data tensors, mask, loss, optimizer) step records of full-batch node tasks; a record the capacity pushes out gives
        # its captured graph back and, if it switched an optimizer to its capturable mode, the optimizer's own settings too
        self.node = TensorKeyedCache(8, on_evict=_retire_record)
        self.graph = None                    # _GraphTaskSteps: one captured training step per graph shape
        self.graph_eval = None               # ... and one captured evaluation step per graph shape

    def __deepcopy__(self, memo):
        return _StepStore()

    def __reduce__(self):
        return (_StepStore, ())

    def prepared_optimizers(self):
        found = [e.value.get("prepared") for e in self.node.entries.values()]
        if self.graph is not None:
            found.append(self.graph.prepared)
        return [p for p in found if p is not None]

    def __del__(self):
        # the model is gone (a fold of a cross-validation loop ended): an optimizer that outlives it gets its own settings
        # back, as release_steps() would have done
        try:
            for prepared in self.prepared_optimizers():
                prepared.restore()
        except Exception:
            pass


def _steps_of(model) -> "_StepStore":
    store = model.__dict__.get("_gnan_steps")
    if store is None:
        store = _StepStore()
        object.__setattr__(model, "_gnan_steps", store)      # a plain attribute: not a sub-module, not in the state_dict
    return store


def release_steps(model) -> None:
    """Drop every captured step of ``model`` now (their hipGraphs and private memory pools) and give an optimizer that was
    switched to its capturable mode its own settings back.  Dropping the model does the same once it is collected."""
    store = model.__dict__.pop("_gnan_steps", None)
    if store is None:
        return
    recs = [e.value for e in store.node.entries.values()]
    for steps in (store.graph, store.graph_eval):
        if steps is not None:
            steps.release_slots()
    if store.graph is not None:
        recs += list(store.graph.buckets.values())
        store.graph.restore_optimizer()
    if store.graph_eval is not None:
        recs += list(store.graph_eval.buckets.values())
    for rec in recs:
        _drop_step(rec)
        if rec.get("prepared") is not None:                # prepared on the first epoch, never captured (fewer than three epochs)
            rec.pop("prepared").restore()
    store.node.clear()
    store.graph = store.graph_eval = None


def _retire_record(rec) -> None:
    _drop_step(rec)
    prepared = rec.pop("prepared", None)
    if prepared is not None:
        prepared.restore()


def _drop_step(rec) -> None:
    step, rec["step"] = rec.get("step"), None
    inner = getattr(step, "step", step)                       # SlottedGraphStep wraps a GraphedStep
    if inner is not None and hasattr(inner, "release"):
        inner.release()


def _labels_of(data, label_index: int, loss_fn) -> torch.Tensor:
    y = data.y
    labels = y[:, label_index].reshape(-1).float() if y.dim() > 1 else y.reshape(-1)
    if bool((labels == -1).any()):                      # {-1, +1} targets -> {0, 1}
        labels = (labels + 1) / 2
    if type(loss_fn).__name__ == "CrossEntropyLoss":
        labels = labels.long()
    return labels


def _resident(obj, device):
    """``obj.to(device)`` unless its tensors are already there (PyG ``Data.to`` copies unconditionally)."""
    x = getattr(obj, "x", None)
    if torch.is_tensor(x) and x.device == torch.device(device):
        return obj
    return obj.to(device)


def _loss_of(loss_fn, outputs, labels):
    if outputs.dim() == 2 and outputs.shape[-1] == 1:
        return loss_fn(outputs.flatten(), labels.float())
    return loss_fn(outputs, labels)


_LABEL_RANGE = None       # TensorKeyedCache: class labels (object identity + version) -> (min, max) as Python ints


def _fused_kind(loss_fn, outputs, labels=None):
    """The loss as ``gnan_loss_step`` knows it, or None: stock torch then computes it (any other loss, CPU tensors).
    Cross entropy: ``gnan_loss_step`` averages over ALL rows it is given, so class labels outside ``[0, C)`` —
    ``ignore_index`` rows (-100 by default), which torch skips and leaves out of the mean — keep the eager loss; the
    range of a label tensor (the batch's ``y``: conservative for masked tasks) is read once per tensor object and version."""
    if not torch.is_tensor(outputs) or not outputs.is_cuda:
        return None
    from . import _lib
    from .losses import loss_kind
    kind = loss_kind(loss_fn, outputs)
    if kind == _lib.LOSS_CROSS_ENTROPY and labels is not None and labels.numel():
        global _LABEL_RANGE
        if _LABEL_RANGE is None:
            from ._cache import TensorKeyedCache
            _LABEL_RANGE = TensorKeyedCache(32)
        rng = _LABEL_RANGE.get((labels,))
        if rng is None:
            lo_hi = torch.stack([labels.min(), labels.max()]).tolist()
            rng = _LABEL_RANGE.put((labels,), None, (int(lo_hi[0]), int(lo_hi[1])))
        if rng[0] < 0 or rng[1] >= outputs.shape[1]:
            return None
    return kind


def _hits(outputs, labels) -> torch.Tensor:
    """Number of correct predictions as a 0-d tensor: arg-max for multi-class logits, sigmoid > 0.5 otherwise."""
    if outputs.dim() == 2 and outputs.shape[-1] > 1:
        return (outputs.argmax(dim=-1) == labels).sum()
    return ((torch.sigmoid(outputs).reshape(-1) > 0.5) == labels).sum()


def _finish(total_loss, hits, n_batches, n_samples, classify, compute_auc, probas, targets):
    loss = float(total_loss) / n_batches
    if not classify:
        return loss, -1
    auc = -1
    if compute_auc:
        from sklearn.metrics import roc_auc_score
        auc = roc_auc_score(np.concatenate(targets), np.concatenate(probas))
    return loss, float(hits) / n_samples, auc


def _single_resident_batch(loader, device):
    """The loader's one batch if it is a full-batch node task whose tensors already live on ``device``, else None.
    Only plain sequences are looked into: iterating a ``DataLoader`` here would start a second iterator per epoch (worker
    start-up, collation) and draw a base seed from the global RNG the reference's loop does not draw."""
    if not isinstance(loader, (list, tuple)) or len(loader) != 1:
        return None
    data = loader[0]
    x = getattr(data, "x", None)
    dev = torch.device(device)
    if not torch.is_tensor(x) or x.device.type != "cuda" or (dev.index is not None and x.device != dev):
        return None
    return data


_GRAPH_FIELDS = ("node_distances", "normalization_matrix", "gnan_rowptr", "gnan_col", "gnan_code", "gnan_cnt", "edge_index")


def _step_record(model, data, mask_name, label_index, loss_fn, optimizer, classify):
    """The record of this (inputs, mask, loss, optimizer) combination, or None if it cannot be keyed (a loss callable that
    cannot be weakly referenced): the epoch then runs eagerly.  The key holds every tensor the step reads — features, mask,
    labels AND the adjacency (swapping or editing it on the same ``Data`` object must not replay the old graph)."""
    cache = _steps_of(model).node
    graph = getattr(data, "gnan_graph", None)
    src = (data.x, getattr(data, mask_name), data.y) + tuple(
        t if torch.is_tensor(t) else None for t in (getattr(data, f, None) for f in _GRAPH_FIELDS))
    extra = (mask_name, int(label_index), bool(classify), optimizer is None, None if graph is None else id(graph))
    rec = cache.get(src, extra)
    if (rec is None or rec["loss_fn"]() is not loss_fn
            or (optimizer is not None and (rec["optimizer"] is None or rec["optimizer"]() is not optimizer))):
        prepared = None
        if rec is not None:
            # the replaced record's optimizer preparation: carried over when the optimizer is the same object (preparing it
            # again would save the PREPARED state as "its own"), handed back otherwise
            _drop_step(rec)
            prepared = rec.pop("prepared", None)
            if prepared is not None and (optimizer is None or rec["optimizer"] is None or rec["optimizer"]() is not optimizer):
                prepared.restore()
                prepared = None
        try:
            loss_ref = weakref.ref(loss_fn)
        except TypeError:
            if prepared is not None:
                prepared.restore()
            return None
        rec = cache.put(src, extra, {"calls": 0, "step": None, "dead": False, "loss_fn": loss_ref,
                                     "optimizer": None if optimizer is None else weakref.ref(optimizer),
                                     "graph": graph})
        if prepared is not None:
            rec["prepared"] = prepared
    return rec


def _graphed_epoch(model, data, loss_fn, optimizer, classify, label_index, compute_auc, mask_name):
    """One epoch of a full-batch node task through a captured step; None if this epoch has to run eagerly."""
    from .graphed import CaptureFailed, GraphedStep
    if not GraphedStep.supported(model, optimizer):
        return None
    rec = _step_record(model, data, mask_name, label_index, loss_fn, optimizer, classify)
    if rec is None or rec["dead"]:
        return None
    if rec["step"] is None:
        if rec["calls"] < GRAPH_AFTER:
            if rec["calls"] == 0 and optimizer is not None:
                # the capturable (fused) update from the FIRST epoch on: the arithmetic of the update never changes in the
                # middle of a run (graphed.prepare_optimizer; undone if the capture fails or the step is released)
                from .graphed import prepare_optimizer
                try:
                    rec["prepared"] = prepare_optimizer(optimizer)
                except CaptureFailed:
                    rec["dead"] = True
                    return None
            rec["calls"] += 1
            return None
        labels = _labels_of(data, label_index, loss_fn).to(data.x.device)
        idx = getattr(data, mask_name).nonzero().flatten()           # integer indices: a boolean mask would synchronise
        labels_m = labels[idx]
        # the step's results side by side — guard flag (float32), loss (float32), hit count (int64): ONE device-to-host copy per
        # epoch instead of three (each a synchronisation of its own: 40 us of a 0.25-ms evaluation pass on the arxiv shape)
        report = torch.zeros(16, dtype=torch.uint8, device=data.x.device)
        rec["report"] = (report, report[0:4].view(torch.float32), report[4:8].view(torch.float32), report[8:16].view(torch.int64))
        rec["fused"] = False

        def loss_of(outputs):
            kind = _fused_kind(loss_fn, outputs, data.y)
            if kind is not None:                     # selection, loss, its gradient and the hit count in one launch
                from .losses import loss_step
                rec["fused"] = True
                loss, hits = loss_step(outputs, labels_m, kind, index=idx, want_hits=True, unit_upstream=True,
                                       out_loss=rec["report"][2], out_hits=rec["report"][3])
                return loss, (hits, outputs.detach().index_select(0, idx) if compute_auc else None)
            picked = outputs.index_select(0, idx)
            loss = _loss_of(loss_fn, picked, labels_m)
            hits = _hits(picked.detach(), labels_m) if classify else None
            return loss, (hits, picked.detach())
        try:
            rec["step"] = GraphedStep(model, data, loss_of, optimizer, warmup=0, prepared=rec.get("prepared"),
                                      guard=rec["report"][1])
            rec["labels"] = labels_m
        except CaptureFailed as e:
            rec["dead"] = True
            if rec.get("prepared") is not None:
                rec.pop("prepared").restore()                  # the eager loop goes on with the caller's own optimizer settings
            warnings.warn(f"gnan_amd: the step could not be captured into a hipGraph ({e}); staying on the eager loop")
            return None
    seen = {}

    def tripped():                        # the one copy: guard | loss | hits
        raw = rec["report"][0].cpu()
        seen["loss"], seen["hits"] = float(raw[4:8].view(torch.float32)), int(raw[8:16].view(torch.int64))
        return bool(float(raw[0:4].view(torch.float32)))
    got = rec["step"].replay(tripped if rec.get("fused") else None)
    if got is None:                       # parameters moved, hyper-parameters changed or the tables outgrew the capture
        rec["step"].release(restore_optimizer=False)           # (the optimizer stays prepared: the step is captured again)
        rec["step"], rec["calls"] = None, GRAPH_AFTER
        return None
    _, loss, (hits, picked) = got
    if rec.get("fused"):
        if "loss" not in seen:            # (an unguarded step: nothing has been read yet)
            tripped()
        loss, hits = seen["loss"], seen["hits"]
    probas = targets = None
    if compute_auc:
        probas = [torch.sigmoid(picked).reshape(-1).cpu().numpy()]
        targets = [rec["labels"].cpu().numpy()]
    return _finish(loss, hits if classify else 0, 1, int(rec["labels"].numel()), classify, compute_auc, probas, targets)


class _GraphTaskSteps:
    """Captured training steps of a graph-level task, one per graph shape (``graphed.SlottedGraphStep``)."""

    def __init__(self, model, optimizer, loss_fn, classify, device):
        self.model, self.loss_fn, self.classify = weakref.ref(model), weakref.ref(loss_fn), classify
        self.optimizer = (lambda: None) if optimizer is None else weakref.ref(optimizer)     # None: evaluation steps
        self.training = optimizer is not None
        self.buckets = {}
        self.captured = 0
        self.prepared = None
        self.total_loss = torch.zeros((), device=device)
        self.hits = torch.zeros((), device=device)
        self.labels = None
        # graph slots: one captured step per (node tier, hop-code tier) — at most six whatever the graphs' shapes
        self.slots, self.slot_calls, self.slot_dead, self.slot_use_cnt = {}, 0, False, None

    def _loss_closure(self):
        loss_fn, classify = self.loss_fn(), self.classify

        def loss_of(outputs, label):
            kind = _fused_kind(loss_fn, outputs)
            if kind is not None and outputs.shape[0] == label.numel():
                from .losses import loss_step
                loss, _ = loss_step(outputs, label, kind, want_hits=False, loss_sum=self.total_loss,
                                    hits_sum=self.hits if classify else None, unit_upstream=True)
                return loss, None
            loss = _loss_of(loss_fn, outputs, label)
            self.total_loss.add_(loss.detach())
            if classify:
                self.hits.add_(_hits(outputs.detach(), label))
            return loss, None
        return loss_of

    def _slot_mode(self, model):
        from .small_graph import slot_mode
        return slot_mode(model) if SLOT_STEPS else None

    def _slot_step(self, model, graph, data, labels):
        """True: the step ran from the slot capture.  False: this graph is for the other routes."""
        from .graphed import CaptureFailed, SlotGraphStep
        from .small_graph import slot_tier
        use_cnt = self._slot_mode(model)
        if use_cnt is None or self.slot_dead:
            return False
        # (nodes, hop codes) tier: the kernels' one- / two-block builds, and tables / bins no larger than the graphs need
        tier = slot_tier(graph) if data.x.dtype == torch.float32 else None
        if tier is None:
            return False
        slot = self.slots.get(tier)
        if slot is None:
            if self.slot_calls < SLOT_AFTER:
                self.slot_calls += 1
                return False
            try:
                if self.prepared is None and self.training:
                    from .graphed import prepare_optimizer
                    self.prepared = prepare_optimizer(self.optimizer())
                slot = self.slots[tier] = SlotGraphStep(model, self.optimizer(), self._loss_closure(), int(data.x.shape[1]), labels,
                                                        use_cnt, prepared=self.prepared, first=(graph, data.x, labels),
                                                        max_nodes=tier[0], n_codes=tier[1])
                self.captured += 1
            except CaptureFailed as e:
                self.slot_dead = True
                warnings.warn(f"gnan_amd: the graph-task slot step could not be captured into a hipGraph ({e}); per-shape steps instead")
                return False
        if not slot.fits(graph, data.x, labels):
            return False
        if slot.run(graph, data.x, labels) is None:          # stale (parameters moved, mode or hyper-parameters changed): capture anew
            slot.step.release(restore_optimizer=False)
            del self.slots[tier]
            self.slot_calls = SLOT_AFTER
            self.captured -= 1
            return False
        return True

    @property
    def slot(self):
        """The captured slot steps (tests / tools): the one of the smallest tier, or None."""
        return self.slots[min(self.slots)] if self.slots else None

    def release_slots(self) -> None:
        for step in self.slots.values():
            step.step.release(restore_optimizer=False)
        self.slots = {}

    def matches(self, model, optimizer, loss_fn, classify) -> bool:
        return (self.model() is model and self.optimizer() is optimizer and self.loss_fn() is loss_fn
                and self.classify == classify)

    def restore_optimizer(self) -> None:
        if self.prepared is not None:
            self.prepared.restore()
            self.prepared = None

    def labels_of(self, data, label_index, loss_fn):
        """``_labels_of`` without its host synchronisation, cached per label tensor."""
        from ._cache import TensorKeyedCache
        if self.labels is None:
            self.labels = TensorKeyedCache(1 << 16)
        hit = self.labels.get((data.y,), label_index)
        if hit is None:
            y = data.y
            lab = y[:, label_index].reshape(-1).float() if y.dim() > 1 else y.reshape(-1)
            lab = torch.where((lab == -1).any(), (lab + 1) / 2, lab)
            lab = lab.long() if type(loss_fn).__name__ == "CrossEntropyLoss" else lab
            hit = self.labels.put((data.y,), label_index, lab.to(self.total_loss.device))
        return hit

    def step(self, data, labels) -> bool:
        """Run this graph's training step from a captured graph if its shape has one (or can get one now)."""
        from .graphed import CaptureFailed, SlottedGraphStep
        model = self.model()
        graph = model.hop_graph(data)
        if not graph.is_dense:
            return False
        if self._slot_step(model, graph, data, labels):
            return True
        if self._slot_mode(model) is not None and not self.slot_dead and graph.n_rows <= 128 and graph.n_codes <= 64:
            return False                                # (a graph for the slots, not replayed this time: eager; no per-shape capture)
        key = (graph.n_rows, graph.n_cols, graph.n_codes, tuple(data.x.shape), data.x.dtype, tuple(labels.shape), labels.dtype)
        rec = self.buckets.get(key)
        if rec is None:
            rec = self.buckets[key] = {"calls": 0, "step": None, "dead": False}
        if rec["dead"]:
            return False
        if rec["step"] is None:
            if rec["calls"] < GRAPH_AFTER or self.captured >= GRAPH_TASK_MAX_SHAPES:
                rec["calls"] += 1
                return False
            loss_of = self._loss_closure()
            try:
                if self.prepared is None and self.training:
                    from .graphed import prepare_optimizer
                    self.prepared = prepare_optimizer(self.optimizer())
                rec["step"] = SlottedGraphStep(model, self.optimizer(), loss_of, graph, data.x, labels, prepared=self.prepared)
                self.captured += 1
            except CaptureFailed as e:
                rec["dead"] = True
                warnings.warn(f"gnan_amd: the graph-task step could not be captured into a hipGraph ({e}); shape stays eager")
                return False
        if rec["step"].run(graph, data.x, labels) is None:
            rec["step"].step.release(restore_optimizer=False)
            rec["step"], rec["calls"] = None, GRAPH_AFTER
            return False
        return True


def _graph_task_steps(model, optimizer, loss_fn, classify, device):
    store = _steps_of(model)
    slot = "graph" if optimizer is not None else "graph_eval"
    steps = getattr(store, slot)
    if steps is None or not steps.matches(model, optimizer, loss_fn, classify):
        if steps is not None:
            for rec in steps.buckets.values():
                _drop_step(rec)
            steps.release_slots()
            steps.restore_optimizer()
        try:
            steps = _GraphTaskSteps(model, optimizer, loss_fn, classify, device)
        except TypeError:                                   # a loss / optimizer that cannot be weakly referenced: eager loop
            steps = None
        setattr(store, slot, steps)
    return steps


def _run(model, loader, loss_fn, device, optimizer, classify, label_index, compute_auc, mask_name, is_graph_task):
    if GRAPHED_STEPS and not is_graph_task and (optimizer is not None or not torch.is_grad_enabled()):
        data = _single_resident_batch(loader, device)
        if data is not None:
            done = _graphed_epoch(model, data, loss_fn, optimizer, classify, label_index, compute_auc, mask_name)
            if done is not None:
                return done
    total_loss = torch.zeros((), device=device)
    hits = torch.zeros((), device=device)
    n_samples, probas, targets = 0, [], []
    replayer = None
    # graph-level tasks (a different small graph per step): one captured step per graph SHAPE — training steps, and evaluation
    # passes too (test_epoch runs after every training epoch, main.py:176-215; eager, a 30-node graph costs ~15 launches)
    if (GRAPHED_STEPS and is_graph_task and (optimizer is not None or not torch.is_grad_enabled()) and not compute_auc
            and torch.device(device).type == "cuda" and hasattr(model, "hop_graph")):
        from .graphed import GraphedStep
        if GraphedStep.supported(model, optimizer):
            replayer = _graph_task_steps(model, optimizer, loss_fn, classify, device)
            if replayer is not None:
                replayer.total_loss.zero_()
                replayer.hits.zero_()
    for data in loader:
        if replayer is not None and torch.is_tensor(getattr(data, "x", None)) and data.x.is_cuda:
            labels = replayer.labels_of(data, label_index, loss_fn)
            if replayer.step(data, labels):              # this graph's whole step was one replayed hipGraph
                n_samples += len(labels)
                continue
        labels = _labels_of(data, label_index, loss_fn).to(device)
        data = _resident(data, device)
        if optimizer is not None:
            optimizer.zero_grad()
        outputs = model.forward(data)
        if isinstance(outputs, tuple):
            outputs = outputs[0]
        kind = _fused_kind(loss_fn, outputs, data.y)     # (the caller's label tensor: a stable object to remember the range of)
        if kind is not None and (not is_graph_task or outputs.shape[0] == labels.numel()):
            # gnan_loss_step: the mask's rows, the loss, its gradient, the hit count and the running totals in one launch
            from .losses import loss_step
            idx = None
            if not is_graph_task:
                idx = getattr(data, mask_name).nonzero().flatten()
                labels = labels[idx]
            n_samples += len(labels)
            loss, _ = loss_step(outputs, labels, kind, index=idx, want_hits=False, loss_sum=total_loss,
                                hits_sum=hits if classify else None, unit_upstream=True)
            if optimizer is not None:
                loss.backward()
                optimizer.step()
            if compute_auc:
                picked = outputs.detach() if idx is None else outputs.detach().index_select(0, idx)
                probas.append(torch.sigmoid(picked).reshape(-1).cpu().numpy())
                targets.append(labels.detach().cpu().numpy())
            del loss, outputs
            continue
        if not is_graph_task:
            mask = getattr(data, mask_name)
            labels, outputs = labels[mask], outputs[mask]
        n_samples += len(labels)
        loss = _loss_of(loss_fn, outputs, labels)
        if optimizer is not None:
            loss.backward()
            optimizer.step()
        total_loss += loss.detach()
        if classify:
            hits += _hits(outputs.detach(), labels)
        if compute_auc:
            probas.append(torch.sigmoid(outputs.detach()).reshape(-1).cpu().numpy())
            targets.append(labels.detach().cpu().numpy())
        # nothing of this step's autograd graph may outlive it: a capture of the NEXT graph's step would meet its
        # AccumulateGrad nodes, bound to this (default) stream — a cross-stream dependency inside a capture (a crash on ROCm)
        del loss, outputs
    if replayer is not None:
        total_loss, hits = total_loss + replayer.total_loss, hits + replayer.hits
    return _finish(total_loss, hits, len(loader), n_samples, classify, compute_auc, probas, targets)


def train_epoch(model, dloader, loss_fn, optimizer, device, classify=True, label_index=0, compute_auc=False,
                is_graph_task=True):
    """One optimisation pass — same signature and return value as trainer.py:23-86."""
    return _run(model, dloader, loss_fn, device, optimizer, classify, label_index, compute_auc, "train_mask",
                is_graph_task)


def test_epoch(model, dloader, loss_fn, device, classify=True, label_index=0, compute_auc=False, val_mask=False,
               is_graph_task=True):
    """One evaluation pass — same signature and return value as trainer.py:89-154 (leaves ``model.eval()`` on)."""
    if model.training:               # trainer.py:97; walking the F x L sub-modules again costs 10 ms on the Cora shape
        model.eval()
    with torch.no_grad():
        return _run(model, dloader, loss_fn, device, None, classify, label_index, compute_auc,
                    "val_mask" if val_mask else "test_mask", is_graph_task)
