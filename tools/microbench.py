#!/usr/bin/env python3
"""Secondary measurements on one MI355X (the other BASELINE.json configs and stage splits).  Prints JSON lines."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnan_amd  # noqa: E402
from gnan_amd import HopGraph, _lib, functional, pwl  # noqa: E402
from gnan_amd import synthetic as syn  # noqa: E402
from gnan_amd.functional import feature_mlps, stack_mlps  # noqa: E402
from gnan_amd.models import GNAN, TensorGNAN  # noqa: E402

DEV = "cuda"


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


def redraw(m):
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() == 2:
                torch.nn.init.xavier_normal_(p, gain=1.0)
            else:
                p.normal_(0.0, 0.5)


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def pwl_split():
    N, F = 10_000_000, 64
    m = TensorGNAN(F, 1, 3, hidden_channels=64, device=DEV)
    redraw(m)
    m = m.to(DEV).eval()
    x = syn.block_features(N, F, 0, N, 1, DEV)
    with torch.no_grad():
        st = stack_mlps(m.fs)
        tb = pwl.build_tables(st)
        t_build = timeit(lambda: pwl.build_tables(st))
        t_look = timeit(lambda: functional._fpwl_launch(x, tb, False))
        t_look_sum = timeit(lambda: functional._fpwl_launch(x, tb, True))
    print(json.dumps({"what": "pwl_split_10M_F64", "build_ms": t_build, "lookup_fx_ms": t_look,
                      "lookup_sum_ms": t_look_sum, "pieces_max": tb.max_pieces, "fpg": tb.features_per_group}))


def dense_inputs(ei, n):
    """The reference's two dense matrices (pre_process_datasets.py:104-142) from ``edge_index``, through the product's own
    preprocessing (all-pairs BFS on the device, HopGraph.from_edge_index): nd = 1/(1+hop) or 0, norm = shell size."""
    g = HopGraph.from_edge_index(torch.as_tensor(ei).to(DEV), n)
    code = g.code.long()
    nd = torch.where(code == 255, torch.zeros((), device=DEV), 1.0 / (1.0 + code.float()))
    norm = torch.gather(g.cnt.float(), 1, code.clamp_max(g.n_codes - 1))
    return nd, norm


def dense_graph(n, avg_deg, rng):
    ei = np.stack([rng.integers(0, n, int(n * avg_deg / 2)), rng.integers(0, n, int(n * avg_deg / 2))])
    ei = np.concatenate([ei, ei[::-1]], 1)
    return ei, dense_inputs(ei, n)


def cora_shaped():
    rng = np.random.default_rng(0)
    n, F, C = 2708, 1434, 7
    ei, (nd, norm) = dense_graph(n, 3.9, rng)
    x = torch.rand(n, F)
    x = x / x.sum(1, keepdim=True)
    x[:, -1] = 1
    d = Bag(x=x.to(DEV), edge_index=torch.from_numpy(ei).to(DEV), node_distances=nd.to(DEV),
            normalization_matrix=norm.to(DEV))
    out = {"what": "cora_shaped_N2708_F1434_C7"}
    for name, cls, kw in [("models.GNAN", GNAN, dict(num_layers=3)), ("models.TensorGNAN", TensorGNAN, dict(n_layers=3))]:
        m = cls(F, C, hidden_channels=64, device=DEV, **kw)
        redraw(m)
        m = m.to(DEV).eval()
        with torch.no_grad():
            out[name + "_fwd_ms"] = timeit(lambda: m.forward(d), reps=5, warm=2)
        opt = torch.optim.SGD(m.parameters(), lr=0.0)     # what trainer.py:66 resets gradients through (not Module.zero_grad,
                                                          # whose walk over 1434 nn.Sequential costs more than the backward)
        def fb():
            opt.zero_grad(set_to_none=True)
            m.forward(d).pow(2).sum().backward()
        out[name + "_fwd_bwd_ms"] = timeit(fb, reps=3, warm=1)
    print(json.dumps(out))


def mutagenicity_shaped():
    rng = np.random.default_rng(0)
    graphs = []
    for _ in range(200):
        n = int(np.clip(round(rng.lognormal(3.3, 0.45)), 4, 417))
        par = np.array([rng.integers(0, i) for i in range(1, n)])
        ei = np.stack([np.arange(1, n), par])
        extra = rng.integers(0, n, (2, max(1, n // 30)))
        ei = np.concatenate([ei, ei[::-1], extra, extra[::-1]], 1)
        nd, norm = dense_inputs(ei, n)
        x = torch.zeros(n, 15)
        x[torch.arange(n), torch.from_numpy(rng.integers(0, 14, n))] = 1
        x[:, -1] = 1
        graphs.append(Bag(x=x.to(DEV), edge_index=torch.from_numpy(ei).to(DEV), node_distances=nd.to(DEV),
                          normalization_matrix=norm.to(DEV)))
    m = TensorGNAN(15, 1, 3, hidden_channels=64, is_graph_task=True, readout_n_layers=0, device=DEV)
    redraw(m)
    m = m.to(DEV).eval()
    opt = torch.optim.SGD(m.parameters(), lr=0.0)

    def fwd_all():
        with torch.no_grad():
            for g in graphs:
                m.forward(g)

    def fb_all():
        for g in graphs:
            opt.zero_grad(set_to_none=True)
            m.forward(g).pow(2).sum().backward()
    a = timeit(fwd_all, reps=3, warm=1)
    b = timeit(fb_all, reps=2, warm=1)
    print(json.dumps({"what": "mutagenicity_shaped_200_graphs", "fwd_ms_per_graph": a[0] / 200,
                      "fwd_bwd_ms_per_graph": b[0] / 200}))


def arxiv_shaped():
    N, E, F = 169_343, 1_166_243, 129
    gen = torch.Generator(device=DEV).manual_seed(0)
    # preferential-attachment-like: destination ~ squared uniform (heavy head)
    src = torch.randint(0, N, (E,), generator=gen, device=DEV)
    dst = (torch.rand(E, generator=gen, device=DEV) ** 3 * N).long().clamp_(0, N - 1)
    g = syn.hop1_csr(src, dst, N)
    x = syn.block_features(N, F, 0, N, 1, DEV)
    d = Bag(x=x, edge_index=None, gnan_graph=g)
    out = {"what": "arxiv_shaped_N169343_E1166243_F129", "max_degree": int((g.rowptr[1:] - g.rowptr[:-1]).max())}
    for C in (1, 40):
        m = TensorGNAN(F, C, 3, hidden_channels=64, device=DEV)
        redraw(m)
        m = m.to(DEV).eval()
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        with torch.no_grad():
            out[f"C{C}_fwd_ms"] = timeit(lambda: m.forward(d), reps=5, warm=2)

        def fb():
            opt.zero_grad(set_to_none=True)
            m.forward(d).pow(2).sum().backward()
        out[f"C{C}_fwd_bwd_ms"] = timeit(fb, reps=3, warm=1)
    print(json.dumps(out))


def arxiv_khop():
    """f-1 at a size the dense matrices cannot reach (2 x 115 GB): K-hop hop-coded CSR built on the device."""
    import time
    from gnan_amd import HopGraph
    N, E, F = 169_343, 1_166_243, 129
    gen = torch.Generator(device=DEV).manual_seed(0)
    src = torch.randint(0, N, (E,), generator=gen, device=DEV)
    dst = (torch.rand(E, generator=gen, device=DEV) ** 3 * N).long().clamp_(0, N - 1)
    ei = torch.stack([src, dst])
    x = syn.block_features(N, F, 0, N, 1, DEV)
    m = TensorGNAN(F, 1, 3, hidden_channels=64, device=DEV)
    redraw(m)
    m = m.to(DEV).eval()
    out = {"what": "arxiv_shaped_khop_preprocessing"}
    for K in (1, 2, 3):
        HopGraph.from_edge_index(ei, N, K, layout="csr" if K > 1 else "auto")          # warm-up (allocator, lazy inits)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = HopGraph.from_edge_index(ei, N, K, layout="csr" if K > 1 else "auto")
        torch.cuda.synchronize()
        out[f"K{K}_build_ms"] = (time.perf_counter() - t0) * 1e3
        out[f"K{K}_pairs"] = g.nnz
        d = Bag(x=x, edge_index=None, gnan_graph=g)
        with torch.no_grad():
            out[f"K{K}_fwd_ms"] = timeit(lambda: m.forward(d), reps=5, warm=2)
    print(json.dumps(out))


if __name__ == "__main__":
    which = sys.argv[1:] or ["pwl", "cora", "muta", "arxiv", "khop"]
    for w in which:
        {"pwl": pwl_split, "cora": cora_shaped, "muta": mutagenicity_shaped, "arxiv": arxiv_shaped,
         "khop": arxiv_khop}[w]()
