#!/usr/bin/env python3
"""gnan_small_graph_fwd on Mutagenicity-sized graphs: device time per launch for a few (n, F, D) shapes."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench as mb  # noqa: E402
from gnan_amd import HopGraph  # noqa: E402


def graph(n, rng):
    tree = np.stack([np.arange(1, n), rng.integers(0, np.arange(1, n))]) if n > 1 else np.zeros((2, 0), dtype=np.int64)
    nd, norm = mb.dense_inputs(np.concatenate([tree, tree[::-1]], axis=1), n)
    return HopGraph.from_dense(nd.cuda(), norm.cuda())


def main():
    rng = np.random.default_rng(0)
    out = {}
    for n, F in ((30, 15), (30, 1), (4, 15), (64, 15), (30, 64)):
        m = mb.TensorGNAN(F, 1, 3, hidden_channels=64, is_graph_task=True, readout_n_layers=0, device="cuda")
        mb.redraw(m)
        m = m.cuda().eval()
        g = graph(n, rng)
        x = torch.rand(n, F, device="cuda")
        d = mb.Bag(x=x, edge_index=None, gnan_graph=g)
        with torch.no_grad():
            for _ in range(5):
                m.forward(d)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            graph_ = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                m.forward(d)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(graph_):
                for _ in range(20):
                    y = m.forward(d)
            graph_.replay()
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(10):
                graph_.replay()
            ev[1].record()
            torch.cuda.synchronize()
        out[f"n{n}_F{F}_D{g.n_codes}"] = round(ev[0].elapsed_time(ev[1]) / 200 * 1e3, 2)
    print(json.dumps({"us_per_forward_replayed": out}))


if __name__ == "__main__":
    main()
