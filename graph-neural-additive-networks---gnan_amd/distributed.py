"""Node-level multi-GPU forward: 1-D vertex partition (edge-cut) with one all-gather per forward.

The reference is single-process (no torch.distributed anywhere); this is the build's own
scaling path for graphs of the RMAT / papers100M shapes.  One process per GPU; rank ``p`` owns the
contiguous node block ``[lo, hi)``: its rows of ``x``, its rows of the hop-coded CSR (global column ids)
and its rows of the output.

    step 1 (local)     operand_p = shape functions on owned rows            (HIP, no communication)
    step 2 (exchange)  operand   = all_gather(operand_p)                    (RCCL over xGMI)
    step 3 (local)     Y_p = rho-weighted aggregation over owned rows       (HIP)

The rest-bucket term needs the column sums of the full operand; every rank computes them from the
gathered operand, so no second collective is needed.  Parameters are replicated.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist


@dataclass
class VertexPartition:
    n_nodes: int
    world: int
    rank: int

    @property
    def block(self) -> int:                      # rows per rank, padded so every shard has the same size
        return -(-self.n_nodes // self.world)

    @property
    def lo(self) -> int:
        return min(self.rank * self.block, self.n_nodes)

    @property
    def hi(self) -> int:
        return min(self.lo + self.block, self.n_nodes)


def _hip_compute() -> Dict[str, Callable]:
    from .functional import column_sums, feature_mlps, rho_aggregate
    return {"feature_mlps": feature_mlps, "column_sums": column_sums, "aggregate": rho_aggregate}


def gather_operand(local: torch.Tensor, part: VertexPartition, group=None) -> torch.Tensor:
    """All-gather the per-node operand rows; returns the ``[n_nodes, W]`` prefix of the padded buffer."""
    if part.world == 1:
        return local
    W = local.shape[1]
    if local.shape[0] != part.block:
        padded = local.new_zeros((part.block, W))
        padded[: local.shape[0]] = local
        local = padded
    full = local.new_empty((part.block * part.world, W))
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full[: part.n_nodes]


def partitioned_forward(x_local: torch.Tensor, graph_local, stacked, lut: torch.Tensor, use_cnt: bool,
                        part: VertexPartition, order: str = "sum_first", out_channels: int = 1, group=None,
                        compute: Optional[Dict[str, Callable]] = None, marks: Optional[Callable] = None):
    """Forward of the node-level path on this rank's block; returns ``out[lo:hi, :out_channels]``.

    ``order='sum_first'`` exchanges the narrow ``[N, C]`` operand (what the drop-in modules do);
    ``order='reference'`` exchanges ``[N, F*C]`` and sums over features after the aggregation
    (the evaluation order of models.py:373-376, which BASELINE's workload is stated in).
    ``compute`` lets the CPU/gloo tests substitute the kernels; the default is the HIP path.
    ``marks(name)`` is called between stages (bench.py records HIP events there).
    """
    ops = compute or _hip_compute()
    mark = marks or (lambda name: None)
    mark("start")
    operand_local = ops["feature_mlps"](x_local, stacked, order == "sum_first")
    mark("fmlp")
    operand = gather_operand(operand_local, part, group)
    mark("gather")
    total = ops["column_sums"](operand)          # rest-bucket operand; every rank derives it from the gathered rows
    mark("total")
    # reference order: the feature sum of models.py:375-376 rides in the aggregation kernel's epilogue
    Y = ops["aggregate"](graph_local, operand, lut, use_cnt, s_total=total,
                         reduce_channels=out_channels if order == "reference" else 0)
    mark("spmm")
    return Y
