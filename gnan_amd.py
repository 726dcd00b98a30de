"""Import shim: the package directory is named ``graph-neural-additive-networks---gnan_amd`` (not a
valid Python identifier), so ``import gnan_amd`` loads it from there under the name ``gnan_amd``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph-neural-additive-networks---gnan_amd")
_spec = importlib.util.spec_from_file_location("gnan_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gnan_amd"] = _mod
_spec.loader.exec_module(_mod)
