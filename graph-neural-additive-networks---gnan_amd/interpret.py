"""What the reference's notebook plots (mutagenicity_visualizations.ipynb cells 4-9):
the distance function rho on the hop grid, every shape function f_k on a value grid, and their product.
All of it falls out of tables the fast path builds anyway; here it is exposed as plain tensors."""
from __future__ import annotations

from typing import Optional

import torch

from .graph import hop_inputs


@torch.no_grad()
def rho_curve(model, max_hop: int) -> torch.Tensor:
    """``rho(1/(1+d))`` for ``d = 0..max_hop`` followed by ``rho(0)`` (unreachable): ``[max_hop + 2, C_rho]``."""
    dev = next(model.rho.parameters()).device
    return model.rho(hop_inputs(max_hop + 2, dev).view(-1, 1))


@torch.no_grad()
def shape_functions(model, grid: torch.Tensor, features: Optional[list] = None) -> torch.Tensor:
    """``f_k(v)`` for every grid value ``v``: ``[len(features), len(grid), C]`` (the notebook evaluates ``f_k(1)``)."""
    ks = range(len(model.fs)) if features is None else features
    dev = next(model.fs[0].parameters()).device
    g = grid.to(dev).float().view(-1, 1)
    return torch.stack([model.fs[k](g) for k in ks], dim=0)


@torch.no_grad()
def contribution_heatmap(model, max_hop: int, value: float = 1.0) -> torch.Tensor:
    """``f_k(value) * rho(1/(1+d))`` — the notebook's feature-by-distance heat map: ``[F, max_hop + 1, C]``."""
    f = shape_functions(model, torch.tensor([value]))[:, 0]          # [F, C]
    r = rho_curve(model, max_hop)[: max_hop + 1]                     # [D, C_rho]
    return f.unsqueeze(1) * r.unsqueeze(0)
