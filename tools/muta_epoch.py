#!/usr/bin/env python3
"""Config 2 at its own size: a 4337-graph Mutagenicity-shaped training epoch (N_g ~ clip(round(LogNormal(3.3,
0.45)), 4, 417), 14 atom types + the ones column, batch_size = 1) through harness.train_epoch / test_epoch on
models.TensorGNAN(is_graph_task=True).  Epochs 1-2 run eagerly (and capture), later epochs replay one hipGraph per graph
shape.  Prints ms per graph by epoch, how many of the epoch's steps were replayed from a captured step (6 kernels: slot
refill, forward, loss, backward, step counters, Adam — tools/graphed_timeline.sh; counted per captured step with
gnan_graph_node_count) and the node counts of those that were not.
  python tools/muta_epoch.py [graphs] [models|standalone|nam]     models: models.TensorGNAN as main.py builds it (post-rho);
  standalone: the stand-alone file's TensorGNAN (pre-rho, GNAN.py:65-67); nam: models.TensorGNAN with a 2-layer NAM read-out."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnan_amd  # noqa: E402,F401
from gnan_amd import HopGraph, harness  # noqa: E402
from gnan_amd import synthetic as syn  # noqa: E402
from gnan_amd.models import TensorGNAN  # noqa: E402

DEV = "cuda"


class Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 4337
    variant = sys.argv[2] if len(sys.argv) > 2 else "models"
    graphs = []
    for ei, x, y in syn.mutagenicity_shaped_graphs(count, seed=0):
        n = x.shape[0]
        hg = HopGraph.from_edge_index(torch.as_tensor(ei).to(DEV), n)
        code = hg.code.long()
        nd = torch.where(code == 255, torch.zeros((), device=DEV), 1.0 / (1.0 + code.float()))
        norm = torch.gather(hg.cnt.float(), 1, code.clamp_max(hg.n_codes - 1))
        graphs.append(Data(x=x.to(DEV), y=torch.tensor([[y]], device=DEV), edge_index=None, node_distances=nd,
                           normalization_matrix=norm))
    sizes = np.array([g.x.shape[0] for g in graphs])
    torch.manual_seed(0)
    if variant == "standalone":
        from gnan_amd.GNAN import TensorGNAN as Standalone
        m = Standalone(15, 1, 3, hidden_channels=64, is_graph_task=True, device=DEV)
    else:
        m = TensorGNAN(15, 1, 3, hidden_channels=64, is_graph_task=True, readout_n_layers=2 if variant == "nam" else 0, device=DEV)
    with torch.no_grad():
        for _, p in m.named_parameters():
            torch.nn.init.xavier_normal_(p, gain=1.0) if p.dim() == 2 else p.normal_(0.0, 0.5)
    m = m.to(DEV)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    out = {"what": "muta_shaped_epoch", "variant": variant, "graphs": count, "mean_nodes": float(sizes.mean()), "max_nodes": int(sizes.max()),
           "over_64_nodes": int((sizes > 64).sum()), "over_128_nodes": int((sizes > 128).sum())}
    train_ms, eval_ms = [], []
    for epoch in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ret = harness.train_epoch(m, graphs, loss_fn, opt, DEV, classify=True, is_graph_task=True)
        torch.cuda.synchronize()
        train_ms.append((time.perf_counter() - t0) / count * 1e3)
        t0 = time.perf_counter()
        harness.test_epoch(m, graphs, loss_fn, DEV, classify=True, val_mask=True, is_graph_task=True)
        torch.cuda.synchronize()
        eval_ms.append((time.perf_counter() - t0) / count * 1e3)
    st = harness._steps_of(m).graph
    slots = st.slots
    replayed = six = 0
    eager_sizes, kernel_hist = [], {}
    for g in graphs:
        hop = m.hop_graph(g)
        key = (hop.n_rows, hop.n_cols, hop.n_codes, tuple(g.x.shape), g.x.dtype, (1,), torch.float32)
        rec = st.buckets.get(key)
        from gnan_amd.small_graph import SLOT_CODE_TIERS
        slot = slots.get((64 if hop.n_rows <= 64 else 128, next((c for c in SLOT_CODE_TIERS if hop.n_codes <= c), 64)))
        if slot is not None and slot.fits(hop, g.x, slot.label):       # the slot step of the graph's tier (graphed.SlotGraphStep)
            replayed += 1
            kn = int(slot.step.graph.kernel_nodes)
            kernel_hist[kn] = kernel_hist.get(kn, 0) + 1
            six += kn <= 6
        elif rec is not None and rec["step"] is not None:
            replayed += 1
            kn = int(rec["step"].step.graph.kernel_nodes)
            kernel_hist[kn] = kernel_hist.get(kn, 0) + 1
            six += kn <= 6
        else:
            eager_sizes.append(int(g.x.shape[0]))
    out.update(train_ms_per_graph_by_epoch=[round(t, 4) for t in train_ms], eval_ms_per_graph_by_epoch=[round(t, 4) for t in eval_ms],
               epoch_s=round(train_ms[-1] * count / 1e3, 3), last=[float(v) for v in ret[:2]], shapes=len(st.buckets),
               captured_shapes=sum(r["step"] is not None for r in st.buckets.values()) + len(slots),
               slot_steps=[list(t) for t in sorted(slots)], slot_replays=sum(int(sl.step.graph.replays) for sl in slots.values()),
               steps_replayed_from_a_captured_step=replayed, share_replayed=round(replayed / count, 4),
               steps_of_at_most_six_kernels=six, share_at_most_six_kernels=round(six / count, 4),
               kernels_per_step_histogram={str(k): v for k, v in sorted(kernel_hist.items())},
               eager_step_node_counts=sorted(eager_sizes)[-20:], reserved_GB=round(torch.cuda.memory_reserved() / 2 ** 30, 2))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
