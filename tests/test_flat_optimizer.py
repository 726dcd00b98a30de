"""graphed.FlatAdamStep on the CPU: the fused Adam / AdamW update over the FlatMLPStore buffers == optimizer.step() over the
F x L Parameters that are views of them (state re-homed into flat tensors, values kept; the ordinary step, state_dict and
load_state_dict keep working).  The GPU twin (tests/test_gpu_graphed.py) checks the captured steps bit for bit."""
import copy

import pytest
import torch

import gnan_amd  # noqa: F401
from gnan_amd.graphed import FlatAdamStep
from gnan_amd.models import GNAN, TensorGNAN


def _make(kind):
    torch.manual_seed(0)
    if kind == "readout":
        m = TensorGNAN(5, 3, 3, hidden_channels=8, is_graph_task=True, readout_n_layers=2)
    elif kind == "gnan":
        m = GNAN(4, 2, num_layers=2, hidden_channels=8, rho_per_feature=True)
    else:
        m = TensorGNAN(6, 1, 3, hidden_channels=8)
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.5)
    return m


def _stores(m):
    for mod in m.modules():
        if hasattr(mod, "fs") and hasattr(mod, "_stacked"):
            mod._stacked("fs", mod.fs)
        if hasattr(mod, "rho") and hasattr(mod, "_stacked"):
            mod._stacked("rho", [mod.rho])
    return [st for mod in m.modules() for st in getattr(mod, "_stores", {}).values()]


def _give_grads(m, seed):
    g = torch.Generator().manual_seed(seed)
    for st in _stores(m):
        for name, buf in st.buf.items():
            st._on_grad(name, torch.randn(buf.shape, generator=g))


@pytest.mark.parametrize("kind", ["readout", "plain"])
@pytest.mark.parametrize("cls,kw", [(torch.optim.Adam, {"weight_decay": 1e-2}), (torch.optim.AdamW, {"weight_decay": 0.1}),
                                    (torch.optim.Adam, {})])
def test_flat_update_equals_the_optimizer_step(kind, cls, kw):
    a, b = _make(kind), _make(kind)
    oa, ob = (cls(m.parameters(), lr=1e-2, fused=True, **kw) for m in (a, b))
    flat = None
    for it in range(6):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            _give_grads(m, it)
        oa.step()
        if it == 2:                                   # two ordinary steps first: their state is carried over
            flat = FlatAdamStep.build(b, ob)
            assert flat is not None and flat.buffers < len(list(b.parameters()))
        flat.step() if flat is not None else ob.step()
        if it == 4:                                   # the ordinary step keeps working on the views
            for m, o in ((a, oa), (b, ob)):
                o.zero_grad(set_to_none=True)
                _give_grads(m, 100)
            oa.step()
            ob.step()
            assert flat.intact(ob)
    # (the CPU kernel rounds its vector body and its scalar tail differently, and flat tensors move the tails: 1e-6, not
    # bit for bit — on the GPU every element goes through the same code)
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert float((p1 - p2).detach().abs().max()) <= 1e-6 * float(p1.detach().abs().max()), n1
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"] == sb["param_groups"]
    for k in sa["state"]:
        for f in ("step", "exp_avg", "exp_avg_sq"):
            want, got = sa["state"][k][f], sb["state"][k][f]
            assert want.shape == got.shape
            assert float((want - got).abs().max()) <= 1e-6 * float(want.abs().max().clamp_min(1e-30)), (k, f)
    ob.load_state_dict(copy.deepcopy(sa))             # new state tensors: the flat step must not be used any more
    assert not flat.intact(ob)


def test_flat_update_declines_what_it_cannot_reproduce():
    m = _make("plain")
    _give_grads(m, 0)
    ok = torch.optim.Adam(m.parameters(), lr=1e-2, fused=True)
    assert FlatAdamStep.build(m, ok) is not None
    assert FlatAdamStep.build(m, torch.optim.SGD(m.parameters(), lr=1e-2)) is None
    assert FlatAdamStep.build(m, torch.optim.Adam(m.parameters(), lr=1e-2, fused=True, amsgrad=True)) is None
    assert FlatAdamStep.build(m, torch.optim.Adam(m.parameters(), lr=1e-2)) is None                     # not the fused update
    ps = list(m.parameters())
    assert FlatAdamStep.build(m, torch.optim.Adam([{"params": ps[:3]}, {"params": ps[3:], "lr": 1e-3}], fused=True)) is None
    assert FlatAdamStep.build(m, torch.optim.Adam(ps[:-1], lr=1e-2, fused=True)) is None                # a parameter left out
    extra = torch.nn.Parameter(torch.zeros(3))
    assert FlatAdamStep.build(m, torch.optim.Adam(ps + [extra], lr=1e-2, fused=True)) is None           # one from elsewhere
    fresh = _make("gnan")                              # no backward pass yet: no gradient buffers
    _stores(fresh)
    assert FlatAdamStep.build(fresh, torch.optim.Adam(fresh.parameters(), lr=1e-2, fused=True)) is None
    # rho_per_feature: rhos[0 .. F-2] are Parameters no forward ever reads (GNAN.py:108-123, 137) — they lie in no store
    _give_grads(fresh, 0)
    assert FlatAdamStep.build(fresh, torch.optim.Adam(fresh.parameters(), lr=1e-2, fused=True)) is None
