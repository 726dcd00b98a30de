"""Run-level counterpart of the reference's ``main.py`` for the GNAN classes (second half).

``run_exp`` takes the arguments of ``main.py:44-49`` in the same order and walks the same loop (main.py:139-303):

* ``Adam(lr, weight_decay=wd)`` (main.py:141), ``ReduceLROnPlateau(factor 0.9, patience 100, min_lr 1e-8)`` stepped on the
  TRAINING loss after every epoch (main.py:148-154, 166);
* a patience counter on the validation loss (:class:`PatienceCounter`; main.py:16-41, 142, 280), stopping only
  when ``early_stop_flag`` (main.py:284) — and the ``train_loss < loss_thresh`` stop (main.py:281);
* one training pass, one validation pass and one test pass per epoch (main.py:156-164, 270-273) through
  ``gnan_amd.harness`` — plus a test pass whenever a checkpoint is written;
* checkpoints ``models/{unique_run_id}_{data_name}_{model_name}_{seed}_best_val_acc.pt`` / ``_best_val_auc.pt`` /
  ``_best_train_loss.pt`` (main.py:172, 201, 232) holding ``model.state_dict()`` — the reference's key names;
* the model: ``GNAN`` for node tasks, ``TensorGNAN`` for graph tasks, with main.py:76-90's keywords; loss type and output
  width by main.py:343-352 (:func:`loss_and_out_dim`).

Quirks of the reference kept on purpose, because they decide WHEN checkpoints are written and what the epoch
functions are told: ``classify=~is_regression`` is truthy for both values of the flag (main.py:159), and the best-validation-
accuracy tracker is overwritten with the validation LOSS (main.py:200).  Dropped: wandb logging, the baselines, the global
``args`` (``seed_index`` replaces ``args.seed``; ``loss_type`` / ``optimizer_type`` are keyword arguments instead of module
globals).  What it adds: the history it returns, and — because the epoch functions replay a captured step from the third
epoch on — the learning rate the scheduler rewrites reaches the captured Adam update through the optimizer's device-tensor
``lr`` (``graphed.PreparedOptimizer``), with no re-capture: ``tests/test_gpu_harness.py`` pins that against the reference's
loss history (golden 320-321).
"""
from __future__ import annotations

import math
import os
from typing import Callable, Optional

import torch

from . import harness


class PatienceCounter:
    """Early-stop bookkeeping of the run loop (behaviour of main.py:16-41 as run_exp uses it: lower is better): how many
    epochs in a row the watched value stayed strictly above the lowest value seen so far.  An equal value counts as an
    improvement (it restarts the count); ``exhausted`` latches once ``patience`` such epochs have passed."""

    __slots__ = ("patience", "lowest", "bad_epochs", "exhausted")

    def __init__(self, patience: int):
        self.patience = int(patience)
        self.lowest = None
        self.bad_epochs = 0
        self.exhausted = False

    def observe(self, value: float) -> bool:
        if self.lowest is not None and value > self.lowest:
            self.bad_epochs += 1
            self.exhausted = self.exhausted or self.bad_epochs >= self.patience
        else:
            self.lowest, self.bad_epochs = value, 0
        return self.exhausted


class CheckpointRules:
    """When the run loop writes a checkpoint (main.py:167-171, 199-203, 229-235).  Each epoch at most one
    validation-side rule fires (AUC when it is computed, accuracy otherwise) and then the training-loss rule.  The
    accuracy rule's threshold is replaced by the validation LOSS when it fires (main.py:200 stores
    ``val_loss``) — that decides at which epochs ``*_best_val_acc.pt`` is rewritten, so it is kept."""

    def __init__(self, use_auc: bool):
        self.use_auc = bool(use_auc)
        self.val_bar = 0                      # the value the validation-side metric has to beat
        self.lowest_train_loss = math.inf

    def due(self, train_loss, val_loss, val_acc, val_auc):
        """Tags of the checkpoints this epoch's numbers call for, in the order they are written."""
        tags = []
        if self.use_auc:
            if val_auc > self.val_bar:
                self.val_bar = val_auc
                tags.append("best_val_auc")
        elif val_acc > self.val_bar:
            self.val_bar = val_loss
            tags.append("best_val_acc")
        if train_loss < self.lowest_train_loss:
            self.lowest_train_loss = train_loss
            tags.append("best_train_loss")
        return tags


def loss_and_out_dim(num_classes: int, is_regression: bool):
    """main.py:343-352: MSE / 1 output for regression, BCE-with-logits / 1 output for two classes, cross-entropy / one output
    per class otherwise."""
    if is_regression:
        return torch.nn.MSELoss, 1
    if num_classes == 2:
        return torch.nn.BCEWithLogitsLoss, 1
    return torch.nn.CrossEntropyLoss, num_classes


def build_model(num_features, out_dim, n_layers, hidden_channels, dropout, device, rho_per_feature, normalize_m,
                is_graph_task, readout_n_layers=0):
    """main.py:76-90 (``model_name == 'gnan'``): ``GNAN`` for node tasks, ``TensorGNAN`` for graph tasks."""
    from .models import GNAN, TensorGNAN
    if not is_graph_task:
        return GNAN(in_channels=num_features, hidden_channels=hidden_channels, num_layers=n_layers, out_channels=out_dim,
                    dropout=dropout, device=device, rho_per_feature=rho_per_feature, normalize_rho=normalize_m)
    return TensorGNAN(in_channels=num_features, hidden_channels=hidden_channels, n_layers=n_layers, out_channels=out_dim,
                      dropout=dropout, device=device, rho_per_feature=rho_per_feature, normalize_rho=normalize_m,
                      is_graph_task=is_graph_task, readout_n_layers=readout_n_layers)


def run_exp(train_loader, val_loader, test_loader, num_features, seeds, n_layers, early_stop_flag, dropout, model_name,
            num_epochs, wandb_flag, wd, hidden_channels, lr, loss_thresh, data_name, unique_run_id, rho_per_feature,
            normalize_m, is_graph_task, num_classes, out_dim, readout_n_layers=0, is_regression=False,
            processed_data_dir="processed_data", compute_auc=False, patience=100, *, seed_index: int = 0,
            loss_type: Optional[Callable] = None, optimizer_type=torch.optim.Adam, checkpoint_dir: str = "models",
            model: Optional[torch.nn.Module] = None, device=None, log: Callable = print):
    """The loop of main.py:44-309 for ``model_name == 'gnan'``.  Returns one dict per seed run:
    ``{"seed", "epochs": [{"train_loss", "train_acc", "val_loss", "val_acc", "test_loss", "test_acc", "lr"} ...],
    "checkpoints": [(epoch, file name) ...], "stopped": why the loop ended, "test_loss", "test_acc", "model"}``.
    ``model``: a ready module to train instead of a freshly constructed one (tests hand over pinned weights)."""
    if model_name != "gnan":
        raise ValueError("only model_name='gnan' lives on this path (the baselines of models.py:8-256 are out of scope)")
    if wandb_flag:
        raise ValueError("wandb logging is not part of this harness")
    if device is None:                                    # main.py:49-52: cuda when there is one, else the CPU (gnan_amd.cpu_route)
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if loss_type is None:
        loss_type, _ = loss_and_out_dim(num_classes, is_regression)
    classify = True                                       # main.py:159: `~is_regression` is -1 or -2, truthy either way
    runs = []
    for i, seed in enumerate(list(seeds)[seed_index:seed_index + 1]):          # main.py:54
        net = model if model is not None else build_model(num_features, out_dim, n_layers, hidden_channels, dropout, device,
                                                          rho_per_feature, normalize_m, is_graph_task, readout_n_layers)
        net.to(device)
        from . import optim_params
        optimizer = optimizer_type(params=optim_params(net), lr=lr, weight_decay=wd)         # main.py:141, over the flat buffers
        loss = loss_type()
        patience_counter = PatienceCounter(patience)                                            # main.py:142
        rules = CheckpointRules(compute_auc)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.9, patience=100, min_lr=1e-8)   # :148-154
        history, checkpoints, stopped = [], [], "num_epochs"

        def save(tag, epoch):
            os.makedirs(checkpoint_dir, exist_ok=True)
            name = f"{unique_run_id}_{data_name}_{model_name}_{seed}_{tag}.pt"                 # main.py:172, 201, 232
            torch.save(net.state_dict(), os.path.join(checkpoint_dir, name))
            checkpoints.append((epoch, name))

        def test_pass():
            return harness.test_epoch(net, dloader=test_loader, loss_fn=loss, classify=classify, device=device,
                                      compute_auc=compute_auc, is_graph_task=is_graph_task)

        for epoch in range(num_epochs):
            train_loss, train_acc, train_auc = harness.train_epoch(net, dloader=train_loader, loss_fn=loss, optimizer=optimizer,
                                                                   classify=classify, device=device, compute_auc=compute_auc,
                                                                   is_graph_task=is_graph_task)
            val_loss, val_acc, val_auc = harness.test_epoch(net, dloader=val_loader, loss_fn=loss, classify=classify,
                                                            device=device, val_mask=True, compute_auc=compute_auc,
                                                            is_graph_task=is_graph_task)
            scheduler.step(train_loss)                                                          # main.py:166
            for tag in rules.due(train_loss, val_loss, val_acc, val_auc):                       # main.py:167-235
                save(tag, epoch)
                test_pass()
            test_loss, test_acc, test_auc = test_pass()                                         # main.py:270-273
            cur_lr = optimizer.param_groups[0]["lr"]
            history.append({"train_loss": train_loss, "train_acc": train_acc, "val_loss": val_loss, "val_acc": val_acc,
                            "test_loss": test_loss, "test_acc": test_acc, "lr": float(cur_lr)})
            log(f"Epoch: {epoch:03d}, Train Loss: {train_loss:.4f}, Train Acc: {train_acc:.4f}, Val Loss: {val_loss:.4f}, "
                f"Val Acc: {val_acc:.4f} Test Loss: {test_loss:.4f}, Test Acc: {test_acc:.4f}")
            out_of_patience = patience_counter.observe(val_loss)                                # main.py:280
            if train_loss < loss_thresh:                                                        # main.py:281-283
                stopped = f"loss under {loss_thresh} at epoch: {epoch}"
                log(stopped)
                break
            if early_stop_flag and out_of_patience:                                       # main.py:284-286
                stopped = f"early stop at epoch: {epoch}"
                log(stopped)
                break
        test_loss, test_acc, test_auc = test_pass()                                             # main.py:298-303
        log(f"Test Loss: {test_loss:.4f}, Test Acc: {test_acc:.4f}")
        runs.append({"seed": int(seed), "epochs": history, "checkpoints": checkpoints, "stopped": stopped,
                     "test_loss": test_loss, "test_acc": test_acc, "model": net, "optimizer": optimizer})
    return runs
