"""gnan_amd.replay: ``model.forward(data)`` and the backward pass behind it as two hipGraph replays under the reference's
UNCHANGED loop shape (tests/reference_loop.py restates trainer.py:23-154: anomaly mode, zero_grad, forward, mask, loss,
backward, a stock optimizer over ``model.parameters()``).  Same numbers as the eager launches, step for step."""
import copy

import numpy as np
import pytest
import torch

import reference_loop
from oracle import gnan_oracle as O
from test_gpu_graphed import DEV, Bag, _graph_task, _model, _need_gpu, _node_task

pytestmark = pytest.mark.gpu


def _plans(model):
    cache = model.__dict__.get("_replays")
    return [] if cache is None else [e.value["plan"] for e in cache.entries.values() if e.value["plan"] is not None]


@pytest.mark.parametrize("n,F,C,dense,loss", [(3000, 129, 1, False, "BCEWithLogitsLoss"), (3000, 64, 1, False, "BCEWithLogitsLoss"),
                                              (300, 9, 4, True, "CrossEntropyLoss"), (5000, 33, 3, False, "CrossEntropyLoss")])
def test_reference_shaped_epochs_replayed_equal_the_eager_ones(n, F, C, dense, loss, monkeypatch):
    """Two copies of a model walk the reference-shaped loop in lock-step, one with the replay switched off: after EVERY
    epoch (training in anomaly mode + evaluation) losses, hit rates and parameters agree; the replaying copy has captured
    one training plan (forward + backward) and one evaluation plan and replayed them from the third epoch on."""
    _need_gpu()
    from gnan_amd import replay
    data = _node_task(n, F, C, dense)
    loss_fn = getattr(torch.nn, loss)()
    a = _model(F, C)
    b = copy.deepcopy(a)
    oa, ob = torch.optim.Adam(a.parameters(), lr=2e-3), torch.optim.Adam(b.parameters(), lr=2e-3)     # main.py:141
    epochs = 7
    for e in range(epochs):
        monkeypatch.setattr(replay, "REPLAY_FORWARD", False)
        ta = reference_loop.train_epoch(a, [data], loss_fn, oa, DEV, classify=True, is_graph_task=False)
        ea = reference_loop.test_epoch(a, [data], loss_fn, DEV, classify=True, val_mask=True, is_graph_task=False)
        monkeypatch.setattr(replay, "REPLAY_FORWARD", True)
        tb = reference_loop.train_epoch(b, [data], loss_fn, ob, DEV, classify=True, is_graph_task=False)
        eb = reference_loop.test_epoch(b, [data], loss_fn, DEV, classify=True, val_mask=True, is_graph_task=False)
        assert abs(ta[0] - tb[0]) <= 1e-6 * max(1.0, abs(ta[0])) and ta[1] == tb[1], (e, ta, tb)
        assert abs(ea[0] - eb[0]) <= 1e-6 * max(1.0, abs(ea[0])) and ea[1] == eb[1], (e, ea, eb)
        scale = max(float(v.abs().max()) for v in a.state_dict().values())
        for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert float((va - vb).abs().max()) <= 1e-6 * scale, (e, k)
    plans = _plans(b)
    assert sorted(p.grad for p in plans) == [False, True] and not _plans(a)
    train = [p for p in plans if p.grad][0]
    evalp = [p for p in plans if not p.grad][0]
    assert train.fwd.replays == train.bwd.replays == epochs - 2 and evalp.fwd.replays == epochs - 2
    # the per-layer gradients the reference's names expose are the flat buffers' views: alive after a replayed backward
    named = dict(b.named_parameters())
    assert named["fs.0.0.weight"].grad is not None and torch.isfinite(named["fs.0.0.weight"].grad).all()
    c = copy.deepcopy(b)                                 # a model that holds plans can be copied: the copy starts without them
    assert not _plans(c)
    with torch.no_grad():
        assert float((c.forward(data) - b.forward(data)).abs().max()) <= 1e-6 * float(b.forward(data).abs().max())
    replay.release(b)
    assert not _plans(b)


def test_replayed_forward_tells_stale_inputs_moved_parameters_and_double_forwards(monkeypatch):
    _need_gpu()
    from gnan_amd import replay
    data = _node_task(3000, 64, 1, False)
    m = _model(64, 1)
    outs = [m.forward(data).detach().clone() for _ in range(4)]          # eager, eager, capture + replay, replay
    assert all(torch.equal(o, outs[0]) for o in outs)
    plan = _plans(m)[0]
    assert plan.grad and plan.fwd.replays == 2
    # a backward pass after a SECOND forward on the same inputs: the first output's activations are gone — refused, loudly
    y1 = m.forward(data)
    y2 = m.forward(data)
    with pytest.raises(RuntimeError, match="replayed again"):
        y1.sum().backward()
    y2.sum().backward()                                                    # the latest one is fine
    g_replayed = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    monkeypatch.setattr(replay, "REPLAY_FORWARD", False)
    m.forward(data).sum().backward()
    monkeypatch.setattr(replay, "REPLAY_FORWARD", True)
    scale = max(float(v.abs().max()) for v in g_replayed.values())
    for k, p in m.named_parameters():
        assert float((p.grad - g_replayed[k]).abs().max()) <= 1e-6 * scale, k
    # gradients accumulate over two backward passes without zero_grad, as autograd's do
    m.zero_grad()
    m.forward(data).sum().backward()
    m.forward(data).sum().backward()
    for k, p in m.named_parameters():
        assert float((p.grad - 2 * g_replayed[k]).abs().max()) <= 2e-6 * scale, k
    # an in-place edit of the inputs: another version, another record — the old plan is not replayed
    replays = plan.fwd.replays
    data.x.mul_(1.0)
    m.forward(data)
    assert plan.fwd.replays == replays
    # parameters moved to new storage: the plan is dropped and captured anew after its warm-up
    data2 = _node_task(3000, 64, 1, False, seed=1)
    m.zero_grad()                                                          # (standing gradients postpone a capture: it would freeze "assign")
    for _ in range(3):
        m.forward(data2)
    assert any(p.fwd.replays >= 1 for p in _plans(m))
    m.float()                                                              # (a no-op cast keeps storage) ...
    m.to(DEV)
    with torch.no_grad():
        ev = [m.forward(data2).clone() for _ in range(4)]
    assert all(torch.equal(o, ev[0]) for o in ev)


def test_tables_that_outgrow_a_replayed_forward_fall_back_to_the_eager_launches(monkeypatch):
    """The captured look-up is sized like the speculative one and checks the tables its own build produced on the device;
    weights whose tables outgrow those sizes trip the plan's guard: that forward runs eagerly (right numbers), and a new
    plan is captured with the larger tables."""
    _need_gpu()
    from gnan_amd import functional, replay
    monkeypatch.setattr(functional, "PWL_MIN_WORK", 1 << 18)
    monkeypatch.setattr(functional, "PWL_MIN_NODES", 1 << 12)
    data = _node_task(5000, 64, 1, False)
    m = _model(64, 1)
    full = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():                                                  # three of four hidden units off: small tables
        for k, p in m.named_parameters():
            if k.startswith("fs") and (".0." in k or ".2." in k):
                p[8:] = 0.0
    with torch.no_grad():
        for _ in range(4):
            m.forward(data)
    plan = _plans(m)[0]
    assert plan.guarded and plan.fwd.replays == 2
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.copy_(full[k])
        got = m.forward(data).clone()                                      # replay, guard trips, eager forward
        assert plan.out is None                                            # (released)
        monkeypatch.setattr(replay, "REPLAY_FORWARD", False)
        want = m.forward(data)
        monkeypatch.setattr(replay, "REPLAY_FORWARD", True)
        assert torch.equal(got, want)
        again = [m.forward(data).clone() for _ in range(3)]                # captured anew with the larger tables
    assert all(torch.equal(o, want) for o in again)
    assert any(p.fwd.replays >= 1 for p in _plans(m))


@pytest.mark.parametrize("cls", ["models", "standalone_no_norm"])
def test_graph_level_loop_replays_small_graphs_through_slots(cls, monkeypatch):
    """trainer.py's graph-level loop (batch_size = 1: another graph object every step) in its own shape: from the third small
    forward on, every graph of up to 128 nodes is copied into the slots of its tier and the tier's plan — captured over the
    slots — is replayed, forward and backward one launch each; graphs beyond the slots run eagerly.  Epoch returns and
    parameters == a twin with the replay switched off."""
    _need_gpu()
    from gnan_amd import GNAN as standalone
    from gnan_amd import replay
    from gnan_amd.models import TensorGNAN
    F = 15
    graphs = _graph_task(36, F, sizes=[12, 30, 9, 23, 130, 12, 41, 77, 5, 128])
    loss_fn = torch.nn.BCEWithLogitsLoss()
    torch.manual_seed(0)
    if cls == "models":
        a = TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=0, device=DEV)
    else:
        a = standalone.TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, normalize_rho=False, device=DEV)
    with torch.no_grad():
        for _, p in a.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.3)
    a = a.to(DEV).eval()
    b = copy.deepcopy(a)
    oa, ob = torch.optim.Adam(a.parameters(), lr=1e-3), torch.optim.Adam(b.parameters(), lr=1e-3)
    for e in range(3):
        monkeypatch.setattr(replay, "REPLAY_FORWARD", False)
        ta = reference_loop.train_epoch(a, graphs, loss_fn, oa, DEV, classify=True, is_graph_task=True)
        ea = reference_loop.test_epoch(a, graphs, loss_fn, DEV, classify=True, val_mask=True, is_graph_task=True)
        monkeypatch.setattr(replay, "REPLAY_FORWARD", True)
        tb = reference_loop.train_epoch(b, graphs, loss_fn, ob, DEV, classify=True, is_graph_task=True)
        eb = reference_loop.test_epoch(b, graphs, loss_fn, DEV, classify=True, val_mask=True, is_graph_task=True)
        assert abs(ta[0] - tb[0]) <= 1e-5 * max(1.0, abs(ta[0])) and ta[1] == tb[1], (e, ta, tb)
        assert abs(ea[0] - eb[0]) <= 1e-5 * max(1.0, abs(ea[0])) and ea[1] == eb[1], (e, ea, eb)
        scale = max(float(v.abs().max()) for v in a.state_dict().values())
        for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert float((va - vb).abs().max()) <= 1e-5 * scale, (e, k)
    # an evaluation loop may COLLECT its outputs: a replayed forward hands out copies, the next replay rewrites nothing of them
    with torch.no_grad():
        kept = [b.forward(d) for d in graphs[:12]]
        again = [b.forward(d).clone() for d in graphs[:12]]
    assert all(torch.equal(u, v) for u, v in zip(kept, again)) and len({float(u.sum()) for u in kept}) > 6
    book = b.__dict__["_replay_slots"]
    plans = [r["plan"] for r in book.plans.values() if r["plan"] is not None]
    train = [p for p in plans if p.grad]
    assert train and all(p.bwd.replays == p.fwd.replays for p in train)
    n_fit = sum(d.x.shape[0] <= 128 for d in graphs)
    assert sum(p.fwd.replays for p in train) >= 3 * n_fit - 2 - len(train)      # all but the two warm-up forwards (and a capture each)
    assert sum(p.fwd.replays for p in plans if not p.grad) >= 3 * n_fit - len(plans)
    # the 130-node graphs do not fit the slots: each is its own input (the same object every epoch) and gets a plan of its own
    big = [p for p in _plans(b)]
    assert len(big) >= 2 and all(p.fwd.replays >= 1 for p in big)
    assert "_replay_slots" not in a.__dict__
    # a copy of a model that holds plans (main.py-style best-model snapshots: copy.deepcopy) starts without them and works
    c = copy.deepcopy(b)
    assert not c.__dict__["_replay_slots"].plans and not _plans(c)
    with torch.no_grad():
        assert torch.equal(c.forward(graphs[0]), b.forward(graphs[0]))
    import pickle
    assert pickle.loads(pickle.dumps(b.__dict__["_replay_slots"])).plans == {}
    replay.release(b)


@pytest.mark.parametrize("shape", ["dense", "csr_attached"])
def test_fresh_input_tensors_every_step_are_adopted_and_replayed(shape):
    """The reference's real cadence (trainer.py:46: ``data.to(device)`` per step — NEW tensor objects with the same contents):
    after a few steps the module adopts the inputs (private copies), captures over them and replays whenever the fresh tensors
    compare equal — outputs and parameters step for step those of the loop over persistent tensors; a step whose contents
    differ runs eagerly on the caller's tensors and is right too."""
    _need_gpu()
    from gnan_amd import replay
    from gnan_amd.models import TensorGNAN
    rng = np.random.default_rng(5)
    n, F, C = 400, 9, 3
    host = {"x": torch.from_numpy(rng.random((n, F), dtype=np.float32))}
    if shape == "dense":
        ei = np.stack([rng.integers(0, n, 1500), rng.integers(0, n, 1500)])
        nd, norm = O.pre_process_dense(np.concatenate([ei, ei[::-1]], axis=1), n)
        host.update(node_distances=nd, normalization_matrix=norm)
        attached = {}
    else:
        src, dst = torch.from_numpy(rng.integers(0, n, 3000)), torch.from_numpy(rng.integers(0, n, 3000))
        from gnan_amd import synthetic as syn
        g = syn.hop1_csr(src.to(DEV), dst.to(DEV), n)
        attached = {"gnan_graph": g}
    y = torch.from_numpy(rng.integers(0, C, n)).to(DEV)

    def fresh(scale=1.0):
        return Bag(edge_index=None, **{k: (v * scale if k == "x" else v).to(DEV) for k, v in host.items()}, **attached)
    torch.manual_seed(0)
    a = TensorGNAN(F, C, 3, hidden_channels=16, device=DEV)
    with torch.no_grad():
        for _, p in a.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.4)
    a = a.to(DEV).eval()
    b = copy.deepcopy(a)
    oa, ob = torch.optim.Adam(a.parameters(), lr=2e-3), torch.optim.Adam(b.parameters(), lr=2e-3)
    persistent = fresh()
    loss_fn = torch.nn.CrossEntropyLoss()
    for step in range(12):
        scale = 1.5 if step == 9 else 1.0                 # one step with other contents: eager, and right
        data_b = fresh(scale)
        data_a = persistent if scale == 1.0 else fresh(scale)
        outs = []
        for m, o, d in ((a, oa, data_a), (b, ob, data_b)):
            o.zero_grad()
            out = m.forward(d)
            loss_fn(out, y).backward()
            o.step()
            outs.append(out.detach())
        assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * float(outs[0].abs().max()), step
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert float((p - q).abs().max()) <= 1e-5 * max(1e-3, float(p.abs().max())), k
    book = b.__dict__.get("_replay_adopted")
    assert book is not None and any(e["static"] is not None for e in book.entries.values())
    plans = [e.value["plan"] for e in b.__dict__["_replays"].entries.values() if e.value.get("plan") is not None]
    assert plans and max(p.fwd.replays for p in plans) >= 3, "the adopted inputs were never replayed"
    assert sum(e["changes"] for e in book.entries.values()) == 1
    replay.release(b)
    assert "_replay_adopted" not in b.__dict__
