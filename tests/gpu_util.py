"""Helpers for the parity tests: fixture -> gnan_amd module on cuda:0 (or, for the CPU route's tests, on the CPU)."""
import numpy as np
import torch

import gnan_amd  # noqa: F401
from gnan_amd import GNAN as amd_standalone
from gnan_amd import models as amd_models

DEV = "cuda"


class Bag:
    """Duck-typed stand-in for a PyG ``Data`` (fields the reference reads at GNAN.py:56,66,147,161)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def build_module(g, dev=DEV):
    m = g.meta
    F = g.inputs["x"].shape[1]
    kw = dict(in_channels=F, out_channels=m["C"], hidden_channels=m["H"], bias=m["bias"], dropout=0.0, device=dev)
    v = m["variant"]
    if v.startswith("standalone_tensor"):
        mod = amd_standalone.TensorGNAN(n_layers=m["L"], normalize_rho=m["normalize_rho"],
                                        is_graph_task=v.endswith("graph"), **kw)
    elif v.startswith("models_tensor"):
        mod = amd_models.TensorGNAN(n_layers=m["L"], normalize_rho=m["normalize_rho"],
                                    is_graph_task=v.endswith("graph"), rho_per_feature=m["rho_per_feature"],
                                    readout_n_layers=m.get("readout_n_layers", 0), **kw)
    elif v == "standalone_gnan":
        mod = amd_standalone.GNAN(n_layers=m["L"], normalize_rho=m["normalize_rho"],
                                  rho_per_feature=m["rho_per_feature"], **kw)
    elif v == "models_gnan":
        mod = amd_models.GNAN(num_layers=m["L"], normalize_rho=m["normalize_rho"],
                              rho_per_feature=m["rho_per_feature"], **kw)
    elif v == "models_nam":
        mod = amd_models.NAM(num_layers=m["L"], **kw)
    else:
        raise ValueError(v)
    sd = {k: torch.from_numpy(np.array(a)) for k, a in g.sd.items()}
    mod.load_state_dict(sd, strict=True)          # key names and shapes are part of the drop-in contract
    return mod.to(dev).eval()


def device_inputs(g, dev=DEV):
    return Bag(**{k: torch.from_numpy(np.array(v)).to(dev) for k, v in g.inputs.items()})


def call(mod, g, data):
    v = g.meta["variant"]
    if v == "models_nam":
        return mod.forward(data.x)
    if v.endswith("gnan"):
        return mod.forward(data, g.meta.get("node_ids"))
    return mod.forward(data)
