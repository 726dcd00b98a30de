"""Mirror of the reference's stand-alone model file: ``from gnan_amd.GNAN import TensorGNAN, GNAN``."""
from .modules import StandaloneGNAN as GNAN
from .modules import StandaloneTensorGNAN as TensorGNAN

GNAN.__name__ = GNAN.__qualname__ = "GNAN"                    # trainer.py:43 / main.py:111 read the class name
TensorGNAN.__name__ = TensorGNAN.__qualname__ = "TensorGNAN"

__all__ = ["TensorGNAN", "GNAN"]
