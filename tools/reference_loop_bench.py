#!/usr/bin/env python3
"""The modules under the reference-SHAPED loop (tests/reference_loop.py: anomaly mode, zero_grad, forward, loss, backward,
stock Adam over model.parameters(), loss.item() per step) on the three training shapes: Mutagenicity-shaped graphs
(batch_size = 1), arxiv-shaped, Cora-shaped.  One JSON line per shape: ms per step with and without anomaly mode; with
``--profile`` the host profile (cProfile, top 28 by own time) of the anomaly-mode loop goes to stderr."""
import cProfile
import io
import json
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import graphed_step as gs  # noqa: E402
import microbench as mb  # noqa: E402
import reference_loop  # noqa: E402

DEV = "cuda"
PROFILE = "--profile" in sys.argv
FLAT = "--flat" in sys.argv                # the optimizer over gnan_amd.optim_params(model) (the flat buffers) instead of model.parameters()
FRESH = "--fresh-inputs" in sys.argv      # the batches live on the host: data.to(device) builds NEW device tensors every step (trainer.py:46)


class HostBatch:
    """A batch as the reference's loader holds it: pinned host tensors; ``to(device)`` makes fresh device tensors each call."""

    def __init__(self, d):
        self.__dict__.update({k: (v.detach().cpu().pin_memory() if torch.is_tensor(v) else v) for k, v in d.__dict__.items()})

    def to(self, device):
        out = type("Data", (), {})()
        out.__dict__.update({k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})
        out.to = lambda dev: out
        return out


def upload_ms(batches):
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            b.to(DEV)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / len(batches) * 1e3
        best = t if best is None else min(best, t)
    return best


def model_for(F, C, graph_task):
    torch.manual_seed(0)
    m = mb.TensorGNAN(F, C, 3, hidden_channels=64, is_graph_task=graph_task, readout_n_layers=0, device=DEV)
    mb.redraw(m)
    return m.to(DEV).eval()


def measure(name, batches, F, C, graph_task, epochs, extra):
    loss_fn = torch.nn.BCEWithLogitsLoss() if C == 1 else torch.nn.CrossEntropyLoss()
    out = {"what": name, "steps_per_epoch": len(batches), **extra}
    if FRESH:
        host = {}
        for b in batches:
            if id(b) not in host:
                host[id(b)] = HostBatch(b)
        batches = [host[id(b)] for b in batches]
        out["fresh_inputs"] = True
        out["upload_ms_per_step"] = round(upload_ms(batches), 4)
        epochs = max(epochs, 4)
    for tag, anomaly in (("anomaly", True), ("plain", False)):
        m = model_for(F, C, graph_task)
        import gnan_amd
        opt = torch.optim.Adam(gnan_amd.optim_params(m) if FLAT else m.parameters(), lr=1e-3)
        out["optimizer_tensors"] = sum(len(g["params"]) for g in opt.param_groups)
        ts = []
        for e in range(epochs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = reference_loop.train_epoch(m, batches, loss_fn, opt, DEV, classify=True, is_graph_task=graph_task,
                                             detect_anomaly=anomaly)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / len(batches) * 1e3)
        out[tag + "_ms_per_step_by_epoch"] = [round(t, 4) for t in ts]
        out[tag + "_ms_per_step"] = round(min(ts[1:] or ts), 4)
        if FRESH:
            out[tag + "_ms_per_step_without_upload"] = round(min(ts[1:] or ts) - out["upload_ms_per_step"], 4)
        out[tag + "_last"] = [float(v) for v in ret]
        if anomaly and PROFILE:
            pr = cProfile.Profile()
            pr.enable()
            reference_loop.train_epoch(m, batches, loss_fn, opt, DEV, classify=True, is_graph_task=graph_task)
            torch.cuda.synchronize()
            pr.disable()
            buf = io.StringIO()
            pstats.Stats(pr, stream=buf).strip_dirs().sort_stats("tottime").print_stats(28)
            print("=====", name, "anomaly-mode epoch,", len(batches), "steps", file=sys.stderr)
            print("\n".join(l[:170] for l in buf.getvalue().splitlines()[:44]), file=sys.stderr)
    print(json.dumps(out), flush=True)


def phases(name, batches, F, C, graph_task):
    """Host time per phase of the reference-shaped step (no anomaly mode): what the host spends ISSUING each phase (perf_counter
    around it, no synchronisation added — the loop's own .item() calls are the only ones), summed over an epoch."""
    import reference_loop as rl
    loss_fn = torch.nn.BCEWithLogitsLoss() if C == 1 else torch.nn.CrossEntropyLoss()
    m = model_for(F, C, graph_task)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for _ in range(2):
        rl.train_epoch(m, batches, loss_fn, opt, DEV, classify=True, is_graph_task=graph_task, detect_anomaly=False)
    acc = {k: 0.0 for k in ("labels", "zero_grad", "forward", "mask+loss", "backward", "optimizer.step", "loss.item", "hit_count")}
    pc = time.perf_counter
    torch.cuda.synchronize()
    t_epoch = pc()
    for data in batches:
        t = pc(); labels = rl._labels(data, 0, loss_fn).to(DEV); acc["labels"] += pc() - t
        t = pc(); opt.zero_grad(); acc["zero_grad"] += pc() - t
        t = pc(); out = m.forward(data); acc["forward"] += pc() - t
        t = pc()
        if not graph_task:
            labels, out = labels[data.train_mask], out[data.train_mask]
        loss = rl._loss(loss_fn, out, labels); acc["mask+loss"] += pc() - t
        t = pc(); loss.backward(); acc["backward"] += pc() - t
        t = pc(); opt.step(); acc["optimizer.step"] += pc() - t
        t = pc(); loss.item(); acc["loss.item"] += pc() - t
        t = pc(); rl._hit_count(out, labels); acc["hit_count"] += pc() - t
    torch.cuda.synchronize()
    total = (pc() - t_epoch) / len(batches) * 1e3
    print(json.dumps({"what": name + " host ms per step by phase", "total": round(total, 4),
                      **{k: round(v / len(batches) * 1e3, 4) for k, v in acc.items()}}), flush=True)


def node_batches(make):
    d, n, F, C = make()
    g = torch.Generator().manual_seed(1)
    d.y = torch.randint(0, max(C, 2), (n,), generator=g).to(DEV)
    r = torch.rand(n, generator=g)
    d.train_mask, d.val_mask, d.test_mask = (r < 0.6).to(DEV), ((r >= 0.6) & (r < 0.8)).to(DEV), (r >= 0.8).to(DEV)
    return [d] * 20, F, C, n


if __name__ == "__main__":
    if FRESH:
        # the loop's label handling (trainer.py:32-40) then runs on HOST tensors: on the GPU boxes' 256 hardware threads a 169k-element
        # comparison costs 37 ms of OpenMP start-up (tests/conftest.py limits the suite's host side the same way)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["muta", "arxiv", "cora"]
    if "muta" in which:
        graphs = gs.muta_shaped()
        measure("muta_shaped_600_graphs", graphs, 15, 1, True, 4, {"graphs": len(graphs)})
        if not FRESH:
            phases("muta_shaped_600_graphs", graphs, 15, 1, True)
    if "arxiv" in which:
        b, F, C, n = node_batches(lambda: gs.arxiv_shaped(1))
        measure("arxiv_shaped_C1", b, F, C, False, 3, {"nodes": n, "features": F})
        if not FRESH:
            phases("arxiv_shaped_C1", b, F, C, False)
    if "arxiv40" in which:
        b, F, C, n = node_batches(lambda: gs.arxiv_shaped(40))
        measure("arxiv_shaped_C40", b, F, C, False, 3, {"nodes": n, "features": F})
    if "cora" in which:
        b, F, C, n = node_batches(gs.cora_shaped)
        measure("cora_shaped", b, F, C, False, 3, {"nodes": n, "features": F})
        if not FRESH:
            phases("cora_shaped", b, F, C, False)
