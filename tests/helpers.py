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


def _np64(v):
    if v is None:
        return None
    if torch.is_tensor(v):
        return v.detach().cpu().double().numpy()
    return np.asarray(v, dtype=np.float64)


def grad_rule(got, truth64, ref32=None, floor=1e-5):
    """The tolerance rule of SURVEY section 8c applied to a set of parameter gradients as ONE vector: the largest deviation
    from the float64 truth over all parameters, relative to the largest float64 gradient entry, must not exceed
    ``max(floor, the same figure for the float32 reference)``.  ``got`` / ``truth64`` / ``ref32``: ``{name: array or tensor}``
    (a missing or ``None`` entry counts as zeros; ``ref32=None``: no float32 reference, the bound is the floor).  ``ref32`` may
    be a callable returning that dict: it is evaluated only when the build's error exceeds the floor (a float32 oracle pass
    over a whole configuration costs minutes of host time and cannot change a verdict the floor already gives).
    Returns ``(ok, e_build, e_ref, worst_name)``."""
    if callable(ref32):
        ok, e_build, _, worst = grad_rule(got, truth64, None, floor)
        if ok:
            return ok, e_build, float("nan"), worst
        ref32 = ref32()
    truth = {k: _np64(v) for k, v in truth64.items() if v is not None}
    scale = max(float(np.abs(v).max()) for v in truth.values())
    e_build, e_ref, worst = 0.0, 0.0, None
    for k in set(truth) | {k for k, v in got.items() if v is not None}:
        t = truth.get(k)
        g = _np64(got.get(k))
        if t is None:
            t = np.zeros_like(g)
        if g is None:
            g = np.zeros_like(t)
        e = float(np.abs(g.reshape(t.shape) - t).max()) / scale
        if e > e_build:
            e_build, worst = e, k
        if ref32 is not None:
            r = _np64(ref32.get(k))
            r = np.zeros_like(t) if r is None else r
            e_ref = max(e_ref, float(np.abs(r.reshape(t.shape) - t).max()) / scale)
    return e_build <= max(floor, e_ref), e_build, e_ref, worst


def module_grads(mod):
    return {k: p.grad for k, p in mod.named_parameters()}


def oracle_grads(loss_of_params, params, dtype):
    """Gradients of ``loss_of_params(p)`` with the parameters cast to ``dtype`` (float32: the reference's own arithmetic
    restated by the oracle; float64: the truth)."""
    p = {k: v.detach().cpu().to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    loss_of_params(p).backward()
    return {k: v.grad for k, v in p.items()}


# Two results that each lie within the floor (1e-5 of the largest entry) of the same truth differ by at most twice the floor:
# the bound of a comparison between two ROUTES of the build (two kernels, eager against replayed, sliced against unsliced)
# where both are checked against the truth elsewhere, or no float64 truth exists at the size.
TWO_FLOORS = 2e-5


def rule(got, truth64, ref32=None, floor=1e-5):
    """SURVEY 8c for one tensor: ``max|got - truth| / max|truth| <= max(floor, the same figure for the float32 reference)``.
    ``ref32``: the float32 reference's result, a callable returning it (evaluated only when the floor does not already
    decide), or None (the bound is the floor).  Returns ``(ok, e_build, e_ref)``."""
    t = _np64(truth64)
    scale = max(float(np.abs(t).max()), 1e-300)
    e_build = float(np.abs(_np64(got).reshape(t.shape) - t).max()) / scale
    if e_build <= floor or ref32 is None:
        return e_build <= floor, e_build, float("nan")
    r = _np64(ref32() if callable(ref32) else ref32)
    e_ref = float(np.abs(r.reshape(t.shape) - t).max()) / scale
    return e_build <= max(floor, e_ref), e_build, e_ref


def assert_rule(got, truth64, ref32=None, what=None, floor=1e-5):
    ok, e_build, e_ref = rule(got, truth64, ref32, floor)
    assert ok, f"{what}: build {e_build:.3e} vs fp32 reference {e_ref:.3e} (floor {floor:g})"


def assert_grads_rule(got, truth64, ref32=None, what=None, floor=1e-5):
    """:func:`grad_rule` on sequences or dicts of gradients (one vector: relative to the largest float64 entry of all)."""
    def as_dict(v):
        if v is None or callable(v) or isinstance(v, dict):
            return v
        return {i: t for i, t in enumerate(v)}
    r = ref32
    if callable(ref32):
        r = lambda: as_dict(ref32())          # noqa: E731
    ok, e_build, e_ref, where = grad_rule(as_dict(got), as_dict(truth64), as_dict(r) if not callable(r) else r, floor)
    assert ok, f"{what} [{where}]: build {e_build:.3e} vs fp32 reference {e_ref:.3e} (floor {floor:g})"
