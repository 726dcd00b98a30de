"""Shared helpers for the parity tests: fixture -> oracle call, tolerance rule."""
import numpy as np
import torch

from oracle import gnan_oracle as O


def params_from(golden, dtype):
    return {k: torch.from_numpy(np.array(v)).to(dtype) for k, v in golden.sd.items()}


def inputs_from(golden, dtype):
    out = {}
    for k, v in golden.inputs.items():
        t = torch.from_numpy(np.array(v))
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def oracle_forward(golden, dtype, params=None):
    """Evaluate the oracle restatement that matches the fixture's reference variant."""
    m = golden.meta
    p = params_from(golden, dtype) if params is None else params
    i = inputs_from(golden, dtype)
    v = m["variant"]
    if v.startswith("standalone_tensor"):
        return O.tensor_gnan_forward_standalone(i["x"], i["node_distances"], i["normalization_matrix"], p,
                                                m["normalize_rho"], v.endswith("graph"))
    if v.startswith("models_tensor"):
        return O.tensor_gnan_forward_models(i["x"], i["node_distances"], i["normalization_matrix"], p,
                                            m["normalize_rho"], v.endswith("graph"),
                                            m.get("readout_n_layers", 0))
    if v in ("standalone_gnan", "models_gnan"):
        return O.gnan_forward(i["x"], i["node_distances"], i["normalization_matrix"], p,
                              m["normalize_rho"], m.get("node_ids"))
    if v == "models_nam":
        return O.nam_forward(i["x"], p)
    if v == "batched_tensor":
        return O.batched_tensor_gnan_forward(i["x"], i["dist"], i["batch"], p, m["graph"])
    raise ValueError(v)


def tolerance_ok(y, ref32, truth64, floor=1e-5):
    """SURVEY §8c rule: error vs the fp64 reference must not exceed max(floor, the fp32 reference's own)."""
    def as_t(v):
        return v.detach().cpu() if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))
    t = as_t(truth64).double()
    e_build = O.rel_err(as_t(y), t)
    e_ref = O.rel_err(as_t(ref32), t)
    return e_build <= max(floor, e_ref), e_build, e_ref
