#!/usr/bin/env python3
"""How long do the batch kernels take for batches of equal graphs, and how much does ONE large graph cost?  (rocprofv3 --kernel-trace)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gnan_amd  # noqa
from gnan_amd import batched

def make(sizes, rng, F=15, C=8):
    out = []
    for n in sizes:
        # a random tree + a few chords, all-pairs hop counts by BFS on the CPU (small graphs)
        adj = [[] for _ in range(n)]
        for v in range(1, n):
            u = int(rng.integers(0, v)); adj[u].append(v); adj[v].append(u)
        hops = np.full((n, n), -1, dtype=np.float32)
        for s in range(n):
            dist = {s: 0}; q = [s]
            for u in q:
                for w in adj[u]:
                    if w not in dist: dist[w] = dist[u] + 1; q.append(w)
            for k, d in dist.items(): hops[s, k] = d
        out.append((torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda(), torch.from_numpy(hops).cuda(),
                    torch.tensor([int(rng.integers(0, C))]).cuda()))
    return out

def main():
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mod = batched.TensorGNAN(15, C, 2, hidden_channels=16, device="cuda").to("cuda")
    loss_fn = torch.nn.CrossEntropyLoss()
    cases = {"32x30": [30] * 32, "31x30+1x100": [30] * 31 + [100], "31x30+1x128": [30] * 31 + [128],
             "1x100": [100], "1x30": [30], "1x64": [64], "1x65": [65], "1x128": [128]}
    for name, sizes in cases.items():
        x, blocks, y, bv = batched.collate(make(sizes, rng, C=C))
        for _ in range(4):
            mod.zero_grad(set_to_none=True)
            out = mod(x, blocks, bv)
            torch.cuda.synchronize()
            loss_fn(out, y).backward()
            torch.cuda.synchronize()
        print("CASE", name, "D", blocks.n_codes, flush=True)
main()
