#!/usr/bin/env python3
"""Full-batch node-task training epochs through the harness: eager loop vs hipGraph replay (BASELINE config 3 shape and
the Cora shape).  Prints one JSON line per configuration: ms per epoch (train step incl. Adam; evaluation pass)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import microbench as mb  # noqa: E402
from gnan_amd import harness  # noqa: E402

DEV = "cuda"


class Data(mb.Bag):
    def to(self, device):
        return self


def arxiv_shaped(C=1):
    N, E, F = 169_343, 1_166_243, 129
    gen = torch.Generator(device=DEV).manual_seed(0)
    src = torch.randint(0, N, (E,), generator=gen, device=DEV)
    dst = (torch.rand(E, generator=gen, device=DEV) ** 3 * N).long().clamp_(0, N - 1)
    d = Data(x=mb.syn.block_features(N, F, 0, N, 1, DEV), edge_index=None, gnan_graph=mb.syn.hop1_csr(src, dst, N))
    return d, N, F, C


def rmat_shaped(C=1):
    """R-MAT 2M nodes / 20M edges, F = 64: large enough for the degree-sorted walk of narrow operands and their hot rows."""
    N, E, F = 2_000_000, 20_000_000, 64
    src, dst = mb.syn.rmat_edges(21, N, E, seed=0, device=DEV)
    d = Data(x=mb.syn.block_features(N, F, 0, N, 1, DEV), edge_index=None, gnan_graph=mb.syn.hop1_csr(src, dst, N))
    return d, N, F, C


def cora_shaped():
    rng = np.random.default_rng(0)
    n, F, C = 2708, 1434, 7
    ei, (nd, norm) = mb.dense_graph(n, 3.9, rng)
    x = torch.rand(n, F)
    x = x / x.sum(1, keepdim=True)
    x[:, -1] = 1
    return Data(x=x.to(DEV), edge_index=None, node_distances=nd.to(DEV), normalization_matrix=norm.to(DEV)), n, F, C


def run(name, make):
    d, n, F, C = make()
    g = torch.Generator().manual_seed(1)
    d.y = torch.randint(0, max(C, 2), (n,), generator=g).to(DEV)
    r = torch.rand(n, generator=g)
    d.train_mask, d.val_mask, d.test_mask = (r < 0.6).to(DEV), ((r >= 0.6) & (r < 0.8)).to(DEV), (r >= 0.8).to(DEV)
    loss_fn = torch.nn.BCEWithLogitsLoss() if C == 1 else torch.nn.CrossEntropyLoss()
    out = {"what": name, "nodes": n, "features": F, "channels": C}
    for tag, on in (("eager", False), ("graphed", True)):
        harness.GRAPHED_STEPS = on
        torch.manual_seed(0)
        m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device=DEV)
        mb.redraw(m)
        m = m.to(DEV).eval()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        tr = lambda: harness.train_epoch(m, [d], loss_fn, opt, DEV, classify=True, is_graph_task=False)
        te = lambda: harness.test_epoch(m, [d], loss_fn, DEV, classify=True, val_mask=True, is_graph_task=False)
        out[tag + "_train_ms"] = mb.timeit(tr, reps=30, warm=5)
        out[tag + "_eval_ms"] = mb.timeit(te, reps=30, warm=5)
        out[tag + "_last_loss"] = tr()[0]
    print(json.dumps(out), flush=True)


def muta_shaped():
    """Mutagenicity-shaped graph-level task: 600 graphs, N ~ clip(round(LogNormal(3.3, 0.45)), 4, 417), random
    trees + N/30 extra edges, one-hot of 14 atom types + ones, batch_size = 1."""
    rng = np.random.default_rng(0)
    F, graphs = 15, []
    for i in range(600):
        n = int(np.clip(np.round(rng.lognormal(3.3, 0.45)), 4, 417))
        tree = np.stack([np.arange(1, n), rng.integers(0, np.arange(1, n))])
        extra = np.stack([rng.integers(0, n, n // 30 + 1), rng.integers(0, n, n // 30 + 1)])
        ei = np.concatenate([tree, extra], axis=1)
        nd, norm = mb.dense_inputs(np.concatenate([ei, ei[::-1]], axis=1), n)
        x = torch.zeros(n, F)
        x[torch.arange(n), torch.from_numpy(rng.integers(0, F - 1, n))] = 1.0
        x[:, -1] = 1.0
        y = torch.tensor([[1.0 if rng.random() < 0.5 else -1.0]])
        graphs.append(Data(x=x.to(DEV), y=y.to(DEV), edge_index=None, node_distances=nd.to(DEV), normalization_matrix=norm.to(DEV)))
    return graphs


def run_graph_task():
    graphs = muta_shaped()
    loss_fn = torch.nn.BCEWithLogitsLoss()
    out = {"what": "muta_shaped_600_graphs", "graphs": len(graphs)}
    for tag, on in (("eager", False), ("graphed", True)):
        harness.GRAPHED_STEPS = on
        torch.manual_seed(0)
        m = mb.TensorGNAN(15, 1, 3, hidden_channels=64, is_graph_task=True, readout_n_layers=0, device=DEV)
        mb.redraw(m)
        m = m.to(DEV).eval()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        ts = []
        for epoch in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = harness.train_epoch(m, graphs, loss_fn, opt, DEV, classify=True, is_graph_task=True)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / len(graphs) * 1e3)
        out[tag + "_ms_per_graph_by_epoch"] = [round(t, 3) for t in ts]
        out[tag + "_last"] = ret
        te = []
        for epoch in range(4):                               # evaluation passes (trainer.py:89-154, after every training epoch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            harness.test_epoch(m, graphs, loss_fn, DEV, classify=True, val_mask=True, is_graph_task=True)
            torch.cuda.synchronize()
            te.append((time.perf_counter() - t0) / len(graphs) * 1e3)
        out[tag + "_eval_ms_per_graph_by_epoch"] = [round(t, 3) for t in te]
        if on:
            st = harness._steps_of(m).graph
            out["shapes"] = len(st.buckets)
            out["captured"] = sum(r["step"] is not None for r in st.buckets.values())
            out["reserved_GB"] = round(torch.cuda.memory_reserved() / 2**30, 2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["arxiv", "arxiv40", "cora"]
    if "muta" in which:
        run_graph_task()
    if "arxiv" in which:
        run("arxiv_shaped_C1", lambda: arxiv_shaped(1))
    if "arxiv40" in which:
        run("arxiv_shaped_C40", lambda: arxiv_shaped(40))
    if "arxiv172" in which:
        run("arxiv_shaped_C172", lambda: arxiv_shaped(172))
    if "cora" in which:
        run("cora_shaped", cora_shaped)
    if "rmat" in which:
        run("rmat_2M_20M_C1", lambda: rmat_shaped(1))
