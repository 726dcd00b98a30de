"""Caches of values derived from caller-owned tensors (hop-coded graphs, padded feature matrices, scales).

A derived value may be reused only while it still describes the SAME tensor with the SAME contents.  A key made of
``data_ptr()`` and ``_version`` does not say that: the reference's loops upload one graph per step
(trainer.py:46: ``data.to(device)``), the previous graph's tensors are freed first, and the caching allocator hands
the next graph of the same size the same addresses with version 0 — the key would match and the previous graph's
hop codes would be reused.  Entries here hold weak references to the source tensor OBJECTS and match on identity
(``ref() is t``) plus the version counter (which every view shares with its base, so in-place writes through any
alias invalidate); an entry disappears as soon as one of its sources is garbage-collected, so the derived copies
(gigabytes for a padded feature matrix) do not outlive them.
"""
from __future__ import annotations

import weakref
from collections import OrderedDict
from typing import Any, Hashable, Optional, Sequence

import torch


class _Entry:
    __slots__ = ("refs", "versions", "extra", "value")

    def __init__(self, tensors: Sequence[Optional[torch.Tensor]], extra: Hashable, value: Any, on_dead):
        self.refs = [None if t is None else weakref.ref(t, on_dead) for t in tensors]
        self.versions = [None if t is None else t._version for t in tensors]
        self.extra = extra
        self.value = value

    def matches(self, tensors: Sequence[Optional[torch.Tensor]], extra: Hashable) -> bool:
        if extra != self.extra or len(tensors) != len(self.refs):
            return False
        for ref, version, t in zip(self.refs, self.versions, tensors):
            if ref is None or t is None:
                if ref is not None or t is not None:
                    return False
            elif ref() is not t or t._version != version:
                return False
        return True

    def dead(self) -> bool:
        return any(ref is not None and ref() is None for ref in self.refs)


class TensorKeyedCache:
    """Up to ``capacity`` ``(source tensors, extra key) -> value`` entries, least recently used dropped first.
    Looked up by the ids of the source objects (O(1)), confirmed by identity through the weak references."""

    def __init__(self, capacity: int, on_evict=None):
        self.capacity = int(capacity)
        self.entries = OrderedDict()
        self.on_evict = on_evict          # called with the value of an entry the capacity pushes out (not on staleness / death)

    @staticmethod
    def _slot(tensors, extra):
        return (tuple(None if t is None else id(t) for t in tensors), extra)

    def _drop(self, k) -> None:
        e = self.entries.get(k)
        if e is not None and e.dead():
            del self.entries[k]

    def get(self, tensors: Sequence[Optional[torch.Tensor]], extra: Hashable = None):
        k = self._slot(tensors, extra)
        e = self.entries.get(k)
        if e is None:
            return None
        if not e.matches(tensors, extra):          # a recycled id or an in-place write since: stale
            del self.entries[k]
            return None
        self.entries.move_to_end(k)
        return e.value

    def put(self, tensors: Sequence[Optional[torch.Tensor]], extra: Hashable, value: Any):
        k = self._slot(tensors, extra)
        self.entries.pop(k, None)
        while len(self.entries) >= self.capacity:
            _, old = self.entries.popitem(last=False)
            if self.on_evict is not None:
                self.on_evict(old.value)
        self.entries[k] = _Entry(tensors, extra, value, lambda _ref, k=k: self._drop(k))
        return value

    def clear(self) -> None:
        self.entries.clear()

    def __deepcopy__(self, memo):                  # a copied / pickled model starts with an empty cache
        return TensorKeyedCache(self.capacity, self.on_evict)

    def __reduce__(self):
        return (TensorKeyedCache, (self.capacity,))

    def __len__(self) -> int:
        return len(self.entries)
