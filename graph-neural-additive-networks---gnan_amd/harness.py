"""Epoch loops with the call contract of the reference's ``trainer.py``.

``train_epoch`` / ``test_epoch`` take the same arguments and return the same tuples as trainer.py:23-86 /
:89-154, so ``main.py``-style drivers can switch by changing one import.  What they reproduce: label handling
(trainer.py:32-40), the node-task mask applied AFTER the forward (:53-55, :125-131), the flatten rule of the
loss (:61-64), accuracy (:5-20) and ROC-AUC (:78,147), the fact that ``test_epoch`` leaves the model in eval
mode (:97).  What they do differently, for the GPU: batches that already live on ``device`` are not copied
again (the reference re-uploads both N x N matrices on every call, trainer.py:46,109), losses and hit counts
are accumulated on the device and read back once per epoch instead of one ``.item()`` sync per batch, and the
autograd anomaly mode that wraps every reference epoch (trainer.py:24) is not switched on.
"""
from __future__ import annotations

import numpy as np
import torch


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


def _run(model, loader, loss_fn, device, optimizer, classify, label_index, compute_auc, mask_name, is_graph_task):
    total_loss = torch.zeros((), device=device)
    hits = torch.zeros((), device=device)
    n_samples, probas, targets = 0, [], []
    for data in loader:
        labels = _labels_of(data, label_index, loss_fn).to(device)
        data = _resident(data, device)
        if optimizer is not None:
            optimizer.zero_grad()
        outputs = model.forward(data)
        if isinstance(outputs, tuple):
            outputs = outputs[0]
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
    return _finish(total_loss, hits, len(loader), n_samples, classify, compute_auc, probas, targets)


def train_epoch(model, dloader, loss_fn, optimizer, device, classify=True, label_index=0, compute_auc=False,
                is_graph_task=True):
    """One optimisation pass — same signature and return value as trainer.py:23-86."""
    return _run(model, dloader, loss_fn, device, optimizer, classify, label_index, compute_auc, "train_mask",
                is_graph_task)


def test_epoch(model, dloader, loss_fn, device, classify=True, label_index=0, compute_auc=False, val_mask=False,
               is_graph_task=True):
    """One evaluation pass — same signature and return value as trainer.py:89-154 (leaves ``model.eval()`` on)."""
    model.eval()
    with torch.no_grad():
        return _run(model, dloader, loss_fn, device, None, classify, label_index, compute_auc,
                    "val_mask" if val_mask else "test_mask", is_graph_task)
