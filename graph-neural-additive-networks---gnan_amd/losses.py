"""The epoch loops' loss step (``harness.py``; trainer.py:44-66): selection of the masked rows, loss, gradient, hit count and
running totals in one launch (``csrc/loss.hip``)."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib
from . import functional as Fn

def loss_kind(loss_fn, outputs: torch.Tensor) -> Optional[int]:
    """``_lib.LOSS_*`` if ``loss_fn(outputs-as-the-trainer-feeds-them, labels)`` (trainer.py:61-64) is one of the two losses
    ``gnan_loss_step`` computes, with the default options it assumes; None otherwise (the caller keeps ``loss_fn``)."""
    if not FUSED_LOSS or not outputs.is_cuda or outputs.dim() != 2 or outputs.dtype != torch.float32:
        return None
    if (type(loss_fn) is torch.nn.BCEWithLogitsLoss and outputs.shape[1] == 1 and loss_fn.weight is None
            and loss_fn.pos_weight is None and loss_fn.reduction == "mean"):
        return _lib.LOSS_BCE_LOGITS
    if (type(loss_fn) is torch.nn.CrossEntropyLoss and outputs.shape[1] > 1 and loss_fn.weight is None
            and loss_fn.reduction == "mean" and loss_fn.ignore_index == -100 and loss_fn.label_smoothing == 0.0):
        return _lib.LOSS_CROSS_ENTROPY
    return None


FUSED_LOSS = os.environ.get("GNAN_FUSED_LOSS", "1") != "0"


def _loss_launch(outputs, labels, kind, index, want_hits, want_grad, loss_sum, hits_sum, out_loss=None, out_hits=None,
                 label_flag=None):
    x = Fn._rows(outputs.detach())
    n_rows, C = x.shape
    n = int(labels.numel())
    lab = labels.detach().contiguous()
    lab = lab.float() if kind == _lib.LOSS_BCE_LOGITS else lab.long()
    idx = None if index is None else index.detach().long().contiguous()
    loss = torch.empty((), dtype=torch.float32, device=x.device) if out_loss is None else out_loss.view(())
    hits = None
    if want_hits:
        hits = torch.empty((), dtype=torch.int64, device=x.device) if out_hits is None else out_hits.view(())
    grad = torch.empty((n_rows, C), dtype=torch.float32, device=x.device) if want_grad else None
    need = _lib.lib().gnan_loss_workspace_bytes(n)
    ws = torch.empty(need // 8, dtype=torch.float64, device=x.device) if need else None
    a = _lib.LossArgs(logits=_lib.ptr(x), n_rows=n_rows, C=C, kind=kind, stride=x.stride(0), index=_lib.ptr(idx), n=n,
                      labels=_lib.ptr(lab), loss=_lib.ptr(loss), hits=_lib.ptr(hits), grad=_lib.ptr(grad),
                      grad_stride=0 if grad is None else grad.stride(0), loss_sum=_lib.ptr(loss_sum), hits_sum=_lib.ptr(hits_sum),
                      # a guarded captured step: a replay whose tables outgrew the capture is rolled back and re-run eagerly —
                      # its truncated look-up's loss must not reach the epoch's totals (it is counted by the re-run)
                      skip_sums=_lib.ptr(Fn.CAPTURE_GUARD) if (loss_sum is not None or hits_sum is not None) else None,
                      workspace=_lib.ptr(ws), workspace_bytes=need, label_flag=_lib.ptr(label_flag))
    _lib.check(_lib.lib().gnan_loss_step(a, _lib.stream_of(x)), "gnan_loss_step")
    return loss, hits, grad


class _LossStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, labels, kind, index, want_hits, loss_sum, hits_sum, unit_upstream, out_loss, out_hits, label_flag):
        loss, hits, grad = _loss_launch(outputs, labels, kind, index, want_hits, ctx.needs_input_grad[0], loss_sum, hits_sum,
                                        out_loss, out_hits, label_flag)
        ctx.grad, ctx.unit_upstream = grad, unit_upstream
        if hits is not None:
            ctx.mark_non_differentiable(hits)
        return (loss, hits) if hits is not None else loss

    @staticmethod
    def backward(ctx, g_loss, *unused):
        grad, ctx.grad = ctx.grad, None
        return (grad if ctx.unit_upstream else grad * g_loss), None, None, None, None, None, None, None, None, None, None


def loss_step(outputs: torch.Tensor, labels: torch.Tensor, kind: int, index: Optional[torch.Tensor] = None,
              want_hits: bool = True, loss_sum: Optional[torch.Tensor] = None, hits_sum: Optional[torch.Tensor] = None,
              unit_upstream: bool = False, out_loss: Optional[torch.Tensor] = None, out_hits: Optional[torch.Tensor] = None,
              label_flag: Optional[torch.Tensor] = None):
    """``(loss, hits)`` of the rows ``index`` of ``outputs`` (all rows without it) by ``gnan_loss_step``: the mean loss as a
    0-d tensor that back-propagates into ``outputs`` (its gradient was formed in the same launch), the hit count as a 0-d
    int64 tensor (None unless ``want_hits``); ``loss_sum`` / ``hits_sum`` (0-d float32 device tensors) are added to in place.
    ``unit_upstream``: the caller promises to call ``backward()`` on this very loss (upstream gradient 1, as the epoch loops
    do, trainer.py:66) — the stored gradient is then handed down as it is instead of being multiplied by it (a launch).
    ``out_loss`` / ``out_hits``: one-element float32 / int64 device tensors to write the results into (a caller that wants to
    read several results with ONE device-to-host copy lays them out next to each other).  ``label_flag``: a one-element
    float32 device tensor a cross-entropy row with a class label outside ``[0, C)`` sets to 1 — such rows (torch's
    ``ignore_index``) are averaged over like any other here, so a caller that cannot check the labels on the host (a replayed
    step) reads the flag when it reads its totals."""
    _lib.require_device(outputs, labels)
    if labels.numel() == 0:
        raise ValueError("loss_step: no rows selected (the mean of an empty set)")
    if index is not None and index.numel() != labels.numel():
        raise ValueError("loss_step: index and labels differ in length")
    if index is None and labels.numel() != outputs.shape[0]:
        raise ValueError("loss_step: one label per output row")
    for t in (loss_sum, hits_sum):
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda or t.numel() != 1):
            raise ValueError("loss_step: the running totals are 0-d float32 device tensors")
    if out_loss is not None and (out_loss.dtype != torch.float32 or out_loss.numel() != 1 or not out_loss.is_cuda):
        raise ValueError("loss_step: out_loss is a one-element float32 device tensor")
    if out_hits is not None and (out_hits.dtype != torch.int64 or out_hits.numel() != 1 or not out_hits.is_cuda):
        raise ValueError("loss_step: out_hits is a one-element int64 device tensor")
    if label_flag is not None and (label_flag.dtype != torch.float32 or label_flag.numel() != 1 or not label_flag.is_cuda):
        raise ValueError("loss_step: label_flag is a one-element float32 device tensor")
    got = _LossStep.apply(outputs, labels, kind, index, want_hits, loss_sum, hits_sum, bool(unit_upstream), out_loss, out_hits,
                          label_flag)
    return got if want_hits else (got, None)
