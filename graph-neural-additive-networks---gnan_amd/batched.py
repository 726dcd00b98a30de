"""Mirror of the reference's batched variant: ``from gnan_amd.batched import TensorGNAN`` (batched_pyg_main.py:98-184).

Many graphs per call: ``dist_batch`` is the block-diagonal matrix of raw hop counts the reference's collate
function builds (batched_pyg_main.py:54-91), ``-1`` marking cross-graph / unreachable pairs.  rho acts on the raw
hop count, there is no shell normalisation, masked pairs contribute nothing (batched_pyg_main.py:151-159), and node
outputs are summed per graph (batched_pyg_main.py:173-181).  In shell terms: hop code = the hop count, the rest
bucket (``-1``) carries weight 0, so the same two kernels serve: rho on the handful of distinct hop counts, then
the hop-coded aggregation.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from .functional import rho_aggregate
from .graph import HopGraph
from .modules import _PathBase


def _two_layer(hidden: int, out: int, bias: bool, dropout: float) -> nn.Sequential:
    """Linear(1,H) / ReLU / Dropout / Linear(H,out): keys 0 and 3 (batched_pyg_main.py:117-131)."""
    return nn.Sequential(nn.Linear(1, hidden, bias=bias), nn.ReLU(), nn.Dropout(dropout),
                         nn.Linear(hidden, out, bias=bias))


def hop_graph_from_counts(dist: torch.Tensor) -> HopGraph:
    """Hop-coded CSR of the listed pairs (``dist >= 0``) of a raw hop-count matrix; ``-1`` pairs are simply absent.

    Block-diagonal batches are almost entirely ``-1``, so only the per-graph blocks are kept: O(sum N_g^2) pairs
    instead of (sum N_g)^2.  Integer work on the device (torch index ops); bit-exact."""
    _lib.require_device(dist)
    if dist.dim() != 2 or dist.shape[0] != dist.shape[1]:
        raise ValueError(f"dist_batch must be square, got {tuple(dist.shape)}")
    n = dist.shape[0]
    listed = dist >= 0
    rows, cols = torch.nonzero(listed, as_tuple=True)            # row-major order == CSR order
    hops = dist[rows, cols]
    codes = hops.round()
    max_hop = int(codes.max()) if codes.numel() else 0
    if codes.numel() and (not bool((codes == hops).all()) or max_hop > 254):
        raise _lib.GnanHipError("dist_batch must hold integer hop counts in [0, 254] or -1")
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dist.device)
    rowptr[1:] = torch.cumsum(listed.sum(dim=1), 0)
    if int(rowptr[-1]) < 2 ** 31:
        rowptr = rowptr.to(torch.int32)
    return HopGraph.from_csr(rowptr, cols.to(torch.int32), codes.to(torch.uint8), n_cols=n, n_codes=max_hop + 2)


class TensorGNAN(_PathBase):
    """``TensorGNAN`` of the batched script — constructor batched_pyg_main.py:99-131, forward :133-184.
    The shape functions and rho are always two layers deep there (``n_layers`` is accepted and unused)."""

    def __init__(self, in_channels, out_channels, n_layers, hidden_channels=16, device='cpu',
                 bias=True, dropout=0.0, is_graph_task=True):
        super().__init__()
        self.device = device
        self.out_channels = out_channels
        self.is_graph_task = is_graph_task
        self.dropout = dropout
        self.fs = nn.ModuleList(_two_layer(hidden_channels, out_channels, bias, dropout) for _ in range(in_channels))
        self.rho = _two_layer(hidden_channels, out_channels, bias, dropout)
        self._init_caches()

    def forward(self, x_batch, dist_batch, batch_vector):
        self._check_dropout()
        _lib.require_device(x_batch, dist_batch)
        g = self._graph_cache.get((dist_batch,), "counts")
        if g is None:
            g = self._graph_cache.put((dist_batch,), "counts", hop_graph_from_counts(dist_batch))
        S = self._features(x_batch, "fs", self.fs, True)                                    # [N, C]
        hops = torch.arange(g.n_codes - 1, dtype=torch.float32, device=x_batch.device).view(-1, 1)
        if self._dropout_active():
            # rho carries a Dropout here (batched_pyg_main.py:126-131) and the reference draws one mask per PAIR
            # (rho runs on all N*N distances, :151).  A table of rho at the distinct hop counts would share one mask
            # among all pairs of a hop count, so while Dropout is active rho is evaluated per listed pair and the
            # weighted sum runs in torch on the device — the cold path, like the shape functions' (_check_dropout).
            deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
            row_of_pair = torch.repeat_interleave(torch.arange(g.n_rows, device=deg.device), deg)
            w_pair = self.rho(hops[g.code.long()])                                          # [nnz, C], one mask per pair
            Y = torch.zeros_like(S).index_add(0, row_of_pair, w_pair * S[g.col.long()])
        else:
            # rho on the handful of distinct hop counts: one launch of the shape-function kernel (one of gnan_fmlp_bwd in the
            # backward pass) instead of the six framework launches of self.rho(hops), as modules._lut_global
            from .functional import feature_mlps
            listed = feature_mlps(hops, self._stacked("rho", [self.rho]), sum_features=False)
            lut = torch.cat([listed, torch.zeros(1, self.out_channels, device=x_batch.device)], dim=0)
            Y = rho_aggregate(g, S, lut, use_cnt=False, with_rest=False)                    # [N, C]
        if not self.is_graph_task:
            return Y
        n_graphs = int(batch_vector.max()) + 1
        out = torch.zeros(n_graphs, Y.shape[1], device=Y.device, dtype=Y.dtype)
        return out.index_add(0, batch_vector.to(Y.device).long(), Y)                        # batched_pyg_main.py:176-181
