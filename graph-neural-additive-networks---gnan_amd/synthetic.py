"""Synthetic inputs of the shapes BASELINE.json names, generated on the device (no datasets offline).

* :func:`rmat_edges` — Graph500 R-MAT with a fixed random vertex permutation;
* :func:`hop1_csr`   — K = 1 hop-coded CSR (self pair = hop 0, every edge = hop 1) for a row block,
  which is what a 1-hop-truncated ``pre_process`` (pre_process_datasets.py:128-140) would list.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .graph import HopGraph


def rmat_edges(scale: int, n_nodes: int, n_edges: int, seed: int, device,
               abcd=(0.57, 0.19, 0.19, 0.05), chunk: int = 1 << 25) -> Tuple[torch.Tensor, torch.Tensor]:
    """First ``n_edges`` R-MAT edges (duplicates kept) whose permuted endpoints are both ``< n_nodes``."""
    a, b, c, _ = abcd
    gen = torch.Generator(device=device).manual_seed(seed)
    perm = torch.randperm(1 << scale, generator=gen, device=device)
    srcs, dsts, kept = [], [], 0
    while kept < n_edges:
        src = torch.zeros(chunk, dtype=torch.int64, device=device)
        dst = torch.zeros(chunk, dtype=torch.int64, device=device)
        for _ in range(scale):
            r = torch.rand(chunk, generator=gen, device=device)
            src = src * 2 + (r >= a + b)
            dst = dst * 2 + (((r >= a) & (r < a + b)) | (r >= a + b + c))
        src, dst = perm[src], perm[dst]
        m = (src < n_nodes) & (dst < n_nodes)
        srcs.append(src[m])
        dsts.append(dst[m])
        kept += int(m.sum())
    return torch.cat(srcs)[:n_edges], torch.cat(dsts)[:n_edges]


def uniform_edges(n_nodes: int, n_edges: int, seed: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    gen = torch.Generator(device=device).manual_seed(seed)
    return (torch.randint(0, n_nodes, (n_edges,), generator=gen, device=device),
            torch.randint(0, n_nodes, (n_edges,), generator=gen, device=device))


def hop1_csr(src: torch.Tensor, dst: torch.Tensor, n_nodes: int, row_lo: int = 0,
             row_hi: Optional[int] = None) -> HopGraph:
    """Hop-coded CSR of rows ``[row_lo, row_hi)``: pair (i, i) with code 0, every edge i->j with code 1.

    Column ids stay global (they index the all-gathered operand).  ``cnt[i] = (1, deg_i, N - 1 - deg_i)``.
    """
    row_hi = n_nodes if row_hi is None else row_hi
    n_rows = row_hi - row_lo
    dev = src.device
    keep = (src >= row_lo) & (src < row_hi)
    s, d = src[keep] - row_lo, dst[keep]
    rows = torch.cat([torch.arange(n_rows, device=dev), s])
    cols = torch.cat([torch.arange(row_lo, row_hi, device=dev), d])
    code = torch.cat([torch.zeros(n_rows, dtype=torch.uint8, device=dev),
                      torch.ones(s.numel(), dtype=torch.uint8, device=dev)])
    order = torch.argsort(rows, stable=True)
    deg = torch.bincount(rows, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(deg, 0)
    if int(rowptr[-1]) < 2 ** 31:
        rowptr = rowptr.to(torch.int32)
    return HopGraph.from_csr(rowptr, cols[order].to(torch.int32), code[order], n_cols=n_nodes, n_codes=3)


def block_features(n_nodes: int, n_feat: int, row_lo: int, row_hi: int, seed: int, device,
                   block: int = 1 << 20) -> torch.Tensor:
    """``x ~ U[0, 1)`` with the last column 1 (pre_process_datasets.py:127), seeded per 2^20-row block so
    every partitioning of the node range sees the same global matrix."""
    out = torch.empty((row_hi - row_lo, n_feat), dtype=torch.float32, device=device)
    b = row_lo // block
    while b * block < row_hi:
        lo, hi = max(row_lo, b * block), min(row_hi, (b + 1) * block)
        gen = torch.Generator(device=device).manual_seed(seed * 1_000_003 + b)
        full = torch.rand((min(block, n_nodes - b * block), n_feat), generator=gen, device=device)
        out[lo - row_lo:hi - row_lo] = full[lo - b * block:hi - b * block]
        b += 1
    out[:, -1] = 1.0
    return out


def preferential_attachment_edges(n_nodes: int, n_edges: int, seed: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """``n_edges`` directed edges ``src -> dst`` of a preferential-attachment graph (the ogbn-arxiv shape,
    ``datasets.py:273-291``).

    Batagelj-Brandes: nodes arrive in order, edge ``e`` leaves node ``floor(e * N / E)`` (about E/N citations each) and its
    head is the endpoint stored in a uniformly drawn slot of the half-edge list written so far, hence
    ``P(dst = v)`` is proportional to ``deg(v)``.  A slot that holds an earlier edge's head is a reference to that edge;
    the references are resolved by pointer jumping (O(log E) vectorised rounds) instead of the sequential loop.  Node ids
    are then permuted at random, as a real data set's are not sorted by arrival.  Generated on the CPU generator, so the
    graph is the same on every device."""
    gen = torch.Generator().manual_seed(seed)
    e = torch.arange(n_edges, dtype=torch.int64)
    src = (e * n_nodes) // n_edges
    r = (torch.rand(n_edges, generator=gen, dtype=torch.float64) * (2 * e + 1).double()).long().clamp_(max=2 * n_edges - 1)
    r = torch.minimum(r, 2 * e)                                   # slot 2e is the edge's own tail (e = 0: a self loop)
    ptr = torch.where(r % 2 == 1, r // 2, torch.full_like(r, -1))       # the earlier edge whose head this edge copies
    val = src[r // 2].clone()                                     # meaningful where ptr < 0 (an even slot holds a tail)
    while True:
        todo = torch.nonzero(ptr >= 0).flatten()
        if todo.numel() == 0:
            break
        p = ptr[todo]
        done = ptr[p] < 0
        val[todo] = torch.where(done, val[p], val[todo])
        ptr[todo] = torch.where(done, torch.full_like(p, -1), ptr[p])
    perm = torch.randperm(n_nodes, generator=gen)
    return perm[src].to(device), perm[val].to(device)


def mutagenicity_shaped_graphs(count: int, seed: int = 0, n_types: int = 14, max_nodes: int = 417):
    """``count`` small undirected graphs of the Mutagenicity shape: ``N_g = clip(round(LogNormal(3.3,
    0.45)), 4, max_nodes)`` nodes (mean about 30), a random tree plus ``N_g // 30 + 1`` extra edges, one-hot atom type of
    ``n_types`` + the ones column (``pre_process_datasets.py:108``).  Yields ``(edge_index [2, m'] int64 numpy (both directions, coalesced), x [N_g,
    n_types + 1] float32 tensor, label +-1)`` — CPU objects; the caller pre-processes and moves them."""
    import numpy as np
    rng = np.random.default_rng(seed)
    F = n_types + 1
    for _ in range(count):
        n = int(np.clip(np.round(rng.lognormal(3.3, 0.45)), 4, max_nodes))
        tree = np.stack([np.arange(1, n), rng.integers(0, np.arange(1, n))])
        extra = np.stack([rng.integers(0, n, n // 30 + 1), rng.integers(0, n, n // 30 + 1)])
        ei = np.concatenate([tree, extra], axis=1)
        x = torch.zeros(n, F)
        x[torch.arange(n), torch.from_numpy(rng.integers(0, n_types, n))] = 1.0
        x[:, -1] = 1.0
        ei = np.concatenate([ei, ei[::-1]], axis=1)
        # coalesced, no self loops — what a PyG data set holds.  (The reference's COO -> LIL conversion would turn a duplicate
        # edge into a weight-2 edge for its Dijkstra; HopGraph.from_edge_index counts it once.)
        ei = np.unique(ei[:, ei[0] != ei[1]], axis=1)
        yield ei, x, (1.0 if rng.random() < 0.5 else -1.0)
