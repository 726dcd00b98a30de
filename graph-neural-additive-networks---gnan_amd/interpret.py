"""What the reference's notebook plots (mutagenicity_visualizations.ipynb cells 4-9): the distance
function rho on the hop grid, every shape function f_k on a value grid, and their product.

Everything here is read off the tables the fast path evaluates the model with: ``gnan_pwl_build`` tabulates each f_k and
rho EXACTLY (they are ReLU MLPs of a scalar, hence piecewise linear: breakpoints, values, slopes), and the curves are
``gnan_fpwl_fwd`` look-ups of a grid in those tables — the same kernels, the same numbers the forward pass uses.
:func:`shape_function_tables` / :func:`rho_table` hand out the tables themselves: the breakpoints are where a plot of f_k
bends, which a value grid can only approximate.  Device tensors in, device tensors out; no CPU path."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .functional import StackedMLP, _c, _fpwl_launch, feature_mlps, stack_mlps
from .graph import hop_inputs


def _stacked(model, what: str) -> StackedMLP:
    p = model._stacked(what, model.fs if what == "fs" else [model.rho]) if hasattr(model, "_stacked") \
        else stack_mlps(model.fs if what == "fs" else [model.rho])
    _lib.require_device(p.w_last)
    return StackedMLP(*[_c(t) for t in p[:6]], *p[6:])


def _tables(p: StackedMLP):
    from .pwl import build_tables
    return build_tables(p)


@torch.no_grad()
def shape_function_tables(model):
    """The exact piecewise-linear tables of all shape functions (:class:`gnan_amd.pwl.PwlTables`): feature ``k`` owns pieces
    ``off[k] .. off[k+1]-1``; on piece ``i``, ``f_k(x) = val[i] + slope[i] * (x - anchor[i])``.  None if some f_k needs
    more pieces than the tables hold."""
    return _tables(_stacked(model, "fs"))


@torch.no_grad()
def rho_table(model):
    """The exact piecewise-linear table of the distance function rho (one 'feature')."""
    return _tables(_stacked(model, "rho"))


def _curve(p: StackedMLP, x: torch.Tensor) -> torch.Tensor:
    """``[n, F * C]``: every function of ``p`` at its column of ``x [n, F]`` — table look-up, or the shape-function
    kernels when the functions do not fit the tables."""
    t = _tables(p)
    if t is None:
        return feature_mlps(x, p, False)
    return _fpwl_launch(x, t, False)


@torch.no_grad()
def rho_curve(model, max_hop: int) -> torch.Tensor:
    """``rho(1/(1+d))`` for ``d = 0..max_hop`` followed by ``rho(0)`` (unreachable): ``[max_hop + 2, C_rho]``."""
    p = _stacked(model, "rho")
    u = hop_inputs(max_hop + 2, p.w_last.device)
    return _curve(p, u.view(-1, 1).contiguous())


@torch.no_grad()
def shape_functions(model, grid: torch.Tensor, features: Optional[list] = None) -> torch.Tensor:
    """``f_k(v)`` for every grid value ``v``: ``[len(features), len(grid), C]`` (the notebook evaluates ``f_k(1)``)."""
    p = _stacked(model, "fs")
    g = grid.to(p.w_last.device).float().reshape(-1)
    y = _curve(p, g.view(-1, 1).expand(-1, p.F).contiguous()).view(g.numel(), p.F, p.C).permute(1, 0, 2)
    return y.contiguous() if features is None else y[list(features)].contiguous()


@torch.no_grad()
def contribution_heatmap(model, max_hop: int, value: float = 1.0) -> torch.Tensor:
    """``f_k(value) * rho(1/(1+d))`` — the notebook's feature-by-distance heat map: ``[F, max_hop + 1, C]``."""
    f = shape_functions(model, torch.tensor([value]))[:, 0]          # [F, C]
    r = rho_curve(model, max_hop)[: max_hop + 1]                     # [D, C_rho]
    return f.unsqueeze(1) * r.unsqueeze(0)
