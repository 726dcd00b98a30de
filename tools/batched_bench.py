#!/usr/bin/env python3
"""f-2 throughput: Mutagenicity-shaped graphs through gnan_amd.batched.TensorGNAN (batched_pyg_main.py:98-184: two-layer shape
functions, hidden 16, 8 output channels) in batches of 32 / 128 / 512 — forward (one launch per batch) and a training
step (forward + cross-entropy + backward + Adam) — next to the one-graph-per-step loop of the same model (batch size 1).
Prints one JSON line per batch size: graphs per second."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnan_amd  # noqa: E402,F401
from gnan_amd import HopGraph, batched  # noqa: E402
from gnan_amd import synthetic as syn  # noqa: E402

DEV = "cuda"


def graphs(count):
    out = []
    for ei, x, y in syn.mutagenicity_shaped_graphs(count, seed=0):
        n = x.shape[0]
        hg = HopGraph.from_edge_index(torch.as_tensor(ei).to(DEV), n)
        hops = torch.where(hg.code == 255, torch.full((), -1.0, device=DEV), hg.code.float())       # raw hop counts, -1 unreachable
        out.append((x.to(DEV), hops, torch.tensor([1 if y > 0 else 0], device=DEV)))
    return out


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    data = graphs(2048)
    torch.manual_seed(0)
    mod = batched.TensorGNAN(15, 8, 2, hidden_channels=16, device="cuda").to(DEV)
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    opt = torch.optim.Adam(mod.parameters(), lr=1e-3)
    loss_fn = torch.nn.CrossEntropyLoss()
    for bs in (1, 32, 128, 512):
        batches = [batched.collate(data[i:i + bs]) for i in range(0, len(data) - bs + 1, bs)][:64]
        row = {"batch_size": bs, "batches": len(batches), "mean_nodes_per_batch": float(np.mean([b[0].shape[0] for b in batches]))}
        for kernel in (True, False):
            batched.BATCH_KERNEL = kernel
            tag = "one_launch" if kernel else "csr_route"
            mod.eval()

            def fwd():
                with torch.no_grad():
                    for x, blocks, y, bv in batches:
                        mod(x, blocks, bv)

            def step():
                for x, blocks, y, bv in batches:
                    opt.zero_grad(set_to_none=True)
                    loss_fn(mod(x, blocks, bv), y).backward()
                    opt.step()
            t_f = timed(fwd, 5)
            t_s = timed(step, 3)
            row[tag + "_fwd_graphs_per_s"] = round(bs * len(batches) / t_f)
            row[tag + "_train_graphs_per_s"] = round(bs * len(batches) / t_s)
            row[tag + "_fwd_ms_per_batch"] = round(t_f / len(batches) * 1e3, 4)
        # the whole training step replayed from one hipGraph launch over slots (batched.GraphedBatchStep): batches that fit
        batched.BATCH_KERNEL = True
        cap = batched.BATCH_KERNEL_MAX_TOTAL_NODES
        if max(b[0].shape[0] for b in batches) <= cap or bs <= 32:
            x0, b0, y0, _ = batches[0]
            try:
                gs = batched.GraphedBatchStep(mod, opt, loss_fn, x0, b0, y0, node_capacity=cap)
                missed = [0]

                def replayed():
                    for x, blocks, y, bv in batches:
                        if gs.run(x, blocks, y) is None:
                            missed[0] += 1
                            opt.zero_grad(set_to_none=True)
                            loss_fn(mod(x, blocks, bv), y).backward()
                            opt.step()
                t_r = timed(replayed, 3)
                row["replayed_train_graphs_per_s"] = round(bs * len(batches) / t_r)
                row["replayed_train_ms_per_batch"] = round(t_r / len(batches) * 1e3, 4)
                row["replayed_kernels_per_step"] = gs.kernel_nodes
                row["batches_that_did_not_fit_the_slots"] = missed[0] // 6
            except Exception as e:                       # noqa: BLE001  (a bench: report, do not die)
                row["replayed_note"] = f"{type(e).__name__}: {str(e)[:200]}"
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
