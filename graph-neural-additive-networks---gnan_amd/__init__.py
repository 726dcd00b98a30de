"""gnan_amd — MI355X-native implementation of GNAN's distance-weighted additive aggregation path.

Drop-in for the reference's ``TensorGNAN`` / ``GNAN`` / ``NAM`` modules
(``from gnan_amd.models import *`` instead of ``from models import *``;
``from gnan_amd.GNAN import TensorGNAN, GNAN`` instead of ``from GNAN import …``).
The arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI in
``include/gnan_hip.h``; see DESIGN.md and INTEGRATION.md.
"""
from . import GNAN, batched, harness, interpret, models  # noqa: F401  (mirror modules + f-3 / f-4)
from .aggregate import rho_aggregate  # noqa: F401
from .functional import StackedMLP, feature_mlps, stack_mlps  # noqa: F401
from .graph import HopGraph, hop_inputs, shell_counts_csr  # noqa: F401



def optim_params(model):
    """What to hand an optimizer for the cheap step: the model's flat parameter buffers (``model.flat_parameters()``) where it
    has them, else ``model.parameters()``.  ``torch.optim.Adam(gnan_amd.optim_params(model), ...)`` updates a dozen tensors
    instead of the F x L per-layer ones ``model.parameters()`` lists (the nn.Module contract, what main.py:141 passes) and
    steps to the same numbers.  One face per optimizer: never mix these with tensors of ``named_parameters()``."""
    flat = getattr(model, "flat_parameters", None)
    return list(flat()) if callable(flat) else list(model.parameters())


__all__ = ["GNAN", "models", "batched", "HopGraph", "StackedMLP", "feature_mlps", "rho_aggregate", "stack_mlps",
           "hop_inputs", "shell_counts_csr", "optim_params"]
