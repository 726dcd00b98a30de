"""Mirror of the GNAN part of the reference's ``models.py``: ``from gnan_amd.models import *``."""
from .modules import GNAN, NAM, TensorGNAN

__all__ = ["NAM", "TensorGNAN", "GNAN"]
