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
        for _, p in m.named_parameters():
            p.normal_(0, 0.5)
    return m


def _params(m, how):
    """``flat``: ``model.flat_parameters()`` — the flat buffers themselves (``gnan_amd.optim_params``); ``per_layer``: the F x L
    tensors of ``parameters()`` / ``named_parameters()`` that are views of them (what main.py:141 hands its optimizer)."""
    return list(m.flat_parameters()) if how == "flat" else list(m.parameters())


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


@pytest.mark.parametrize("how", ["flat", "per_layer"])
@pytest.mark.parametrize("kind", ["readout", "plain"])
@pytest.mark.parametrize("cls,kw", [(torch.optim.Adam, {"weight_decay": 1e-2}), (torch.optim.AdamW, {"weight_decay": 0.1}),
                                    (torch.optim.Adam, {})])
def test_flat_update_equals_the_optimizer_step(kind, cls, kw, how):
    a, b = _make(kind), _make(kind)
    oa, ob = (cls(_params(m, how), lr=1e-2, fused=True, **kw) for m in (a, b))
    flat = None
    for it in range(6):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            _give_grads(m, it)
        oa.step()
        if it == 2:                                   # two ordinary steps first: their state is carried over
            flat = FlatAdamStep.build(b, ob)
            assert flat is not None and flat.buffers == len(list(b.flat_parameters())) < len(list(b.named_parameters()))
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


@pytest.mark.parametrize("how", ["flat", "per_layer"])
def test_flat_update_declines_what_it_cannot_reproduce(how):
    m = _make("plain")
    _give_grads(m, 0)
    ok = torch.optim.Adam(_params(m, how), lr=1e-2, fused=True)
    assert FlatAdamStep.build(m, ok) is not None
    assert FlatAdamStep.build(m, torch.optim.SGD(_params(m, how), lr=1e-2)) is None
    assert FlatAdamStep.build(m, torch.optim.Adam(_params(m, how), lr=1e-2, fused=True, amsgrad=True)) is None
    assert FlatAdamStep.build(m, torch.optim.Adam(_params(m, how), lr=1e-2)) is None                     # not the fused update
    ps = _params(m, how)
    assert FlatAdamStep.build(m, torch.optim.Adam([{"params": ps[:3]}, {"params": ps[3:], "lr": 1e-3}], fused=True)) is None
    assert FlatAdamStep.build(m, torch.optim.Adam(ps[:-1], lr=1e-2, fused=True)) is None                # a parameter left out
    extra = torch.nn.Parameter(torch.zeros(3))
    assert FlatAdamStep.build(m, torch.optim.Adam(ps + [extra], lr=1e-2, fused=True)) is None           # one from elsewhere
    fresh = _make("gnan")                              # no backward pass yet: no gradient buffers
    _stores(fresh)
    assert FlatAdamStep.build(fresh, torch.optim.Adam(_params(fresh, how), lr=1e-2, fused=True)) is None
    # rho_per_feature: rhos[0 .. F-2] are Parameters no forward ever reads (GNAN.py:108-123, 137) — they lie in no store
    _give_grads(fresh, 0)
    assert FlatAdamStep.build(fresh, torch.optim.Adam(_params(fresh, how), lr=1e-2, fused=True)) is None


def test_parameters_is_torchs_own_and_flat_parameters_are_the_buffers():
    """``model.parameters()`` is the nn.Module contract (the tensors of ``named_parameters()``, same objects, same order);
    ``model.flat_parameters()`` / ``gnan_amd.optim_params(model)``: a dozen flat Parameters (+ the tensors no store holds) with
    the same element count (main.py:92-97 sums ``numel``); names, shapes and ``state_dict`` keys are the per-layer ones; an
    in-place update of a flat Parameter is an update of the per-layer views; a stock optimizer over the flat face steps to the
    same numbers as one over ``parameters()``; ``zero_grad`` of either kind is honoured; ``FLAT_PARAMETERS = True`` is the
    round-5 behaviour (``parameters()`` yields the flat face)."""
    import gnan_amd
    from gnan_amd import modules
    for kind in ("readout", "gnan", "plain"):
        m, twin = _make(kind), _make(kind)
        named = dict(m.named_parameters())
        assert [id(p) for p in m.parameters()] == [id(p) for p in named.values()]            # torch's own
        flat = list(m.flat_parameters())
        assert [id(p) for p in gnan_amd.optim_params(m)] == [id(p) for p in flat]
        assert sum(p.numel() for p in flat) == sum(p.numel() for p in named.values())
        assert len({id(p) for p in flat}) == len(flat) and len(flat) < len(named)
        assert list(m.state_dict().keys()) == list(twin.state_dict().keys())
        with torch.no_grad():
            flat[0].mul_(2.0)
        assert torch.equal(named["fs.0.0.weight"], 2.0 * dict(twin.named_parameters())["fs.0.0.weight"])
        with torch.no_grad():
            flat[0].mul_(0.5)
        oa = torch.optim.Adam(gnan_amd.optim_params(m), lr=1e-2, weight_decay=1e-3)
        ob = torch.optim.Adam(twin.parameters(), lr=1e-2, weight_decay=1e-3)                  # main.py:141
        for it in range(3):
            oa.zero_grad()
            ob.zero_grad()
            _give_grads(m, it)
            _give_grads(twin, it)
            if it == 1:                                    # a second backward pass before the step: gradients add up, both ways
                _give_grads(m, 50)
                _give_grads(twin, 50)
            for (k, p), (_, q) in zip(m.named_parameters(), twin.named_parameters()):
                if k.startswith("rhos.") and not k.startswith(f"rhos.{len(m.fs) - 1}."):
                    assert p.grad is None
                else:
                    assert torch.equal(p.grad, q.grad), k
            oa.step()
            ob.step()
        for (k, p), (_, q) in zip(m.named_parameters(), twin.named_parameters()):
            assert float((p - q).abs().max()) <= 1e-6 * float(q.abs().max()), k
        m.requires_grad_(False)
        assert not any(p.requires_grad for p in m.flat_parameters()) and not any(p.requires_grad for p in m.parameters())
    assert gnan_amd.optim_params(torch.nn.Linear(2, 2)).__len__() == 2                       # any other module: its parameters()
    try:
        modules.FLAT_PARAMETERS = True
        m = _make("plain")
        assert [id(p) for p in m.parameters()] == [id(p) for p in m.flat_parameters()]
    finally:
        modules.FLAT_PARAMETERS = False
