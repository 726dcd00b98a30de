#!/usr/bin/env python3
"""Forward / backward launch times of the one-launch small-graph kernels: post-rho vs pre-rho (with and without the workspace
that lets rho's n x D arguments be split over workgroups)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gnan_amd  # noqa
from gnan_amd import HopGraph, small_graph
from gnan_amd.functional import StackedMLP
from gnan_amd import synthetic as syn

DEV = "cuda"
def main():
    n_target = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    for ei, x, y in syn.mutagenicity_shaped_graphs(200, seed=0):
        if x.shape[0] == n_target:
            break
    n, F = x.shape
    g = HopGraph.from_edge_index(torch.as_tensor(ei).to(DEV), n)
    rng = np.random.default_rng(0)
    H, L = 64, 3
    def mlp(Fk, bias):
        t = lambda *s: torch.from_numpy((rng.standard_normal(s) * 0.5).astype(np.float32)).to(DEV).requires_grad_(True)
        return [t(Fk, H), t(Fk, H) if bias else None, t(1, Fk, H, H), t(1, Fk, H) if bias else None, t(Fk, 1, H), t(Fk, 1) if bias else None]
    fp, rp = mlp(F, True), mlp(1, False)
    f, r = StackedMLP(*fp, L, H, 1, F), StackedMLP(*rp, L, H, 1, 1)
    xs = x.to(DEV)
    out = {"n": n, "F": F, "D": g.n_codes}
    live = [t for t in fp + rp if t is not None]
    for name, mode, ws_on in (("post_rho", True, True), ("pre_rho_split", "pre", True), ("pre_rho_one_group", "pre", False)):
        real = small_graph._workspace
        if not ws_on:
            small_graph._workspace = lambda dev, need: torch.zeros(4, dtype=torch.int32, device=dev) if need > 200000 else real(dev, need)
        fw, bw = [], []
        for it in range(30):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            o = small_graph.small_graph_forward(xs, g, f, r, mode, True)
            e[1].record()
            torch.autograd.grad(o.sum(), live)
            e[2].record()
            torch.cuda.synchronize()
            fw.append(e[0].elapsed_time(e[1])); bw.append(e[1].elapsed_time(e[2]))
        small_graph._workspace = real
        out[name] = {"fwd_ms_min": round(min(fw[5:]), 4), "bwd_ms_min": round(min(bw[5:]), 4)}
    print(json.dumps(out))
main()
