"""hipGraph replay of whole training / evaluation steps (gnan_amd/graphed.py, harness): same numbers as the eager loop."""
import copy
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")


def _node_task(n, F, C, dense, seed=0):
    from gnan_amd import HopGraph, synthetic as syn
    from oracle import gnan_oracle as O
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, F, generator=g)
    x[:, -1] = 1.0
    y = torch.randint(0, max(C, 2), (n,), generator=g)
    masks = torch.rand(n, generator=g)
    data = Bag(x=x.to(DEV), y=y.to(DEV), edge_index=None, train_mask=(masks < 0.6).to(DEV),
               val_mask=((masks >= 0.6) & (masks < 0.8)).to(DEV), test_mask=(masks >= 0.8).to(DEV))
    if dense:
        ei = np.stack([np.random.default_rng(seed).integers(0, n, 3 * n), np.random.default_rng(seed + 1).integers(0, n, 3 * n)])
        nd, norm = O.pre_process_dense(np.concatenate([ei, ei[::-1]], axis=1), n)
        data.node_distances, data.normalization_matrix = nd.to(DEV), norm.to(DEV)
    else:
        src, dst = syn.uniform_edges(n, 6 * n, seed, DEV)
        data.gnan_graph = syn.hop1_csr(src, dst, n)
    return data


def _model(F, C, seed=0):
    from gnan_amd.models import TensorGNAN
    torch.manual_seed(seed)
    m = TensorGNAN(F, C, 3, hidden_channels=32, device=DEV)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() == 2:
                torch.nn.init.xavier_normal_(p, gain=1.0)
            else:
                p.normal_(0.0, 0.3)
    return m.to(DEV).eval()


@pytest.mark.parametrize("n,F,C,dense,loss", [(3000, 129, 1, False, "BCEWithLogitsLoss"), (300, 9, 4, True, "CrossEntropyLoss"),
                                              (2500, 120, 3, False, "CrossEntropyLoss")])
def test_graphed_epochs_match_eager_epochs(n, F, C, dense, loss, monkeypatch):
    """Twenty-four epochs of harness.train_epoch / test_epoch on a full-batch node task: with hipGraph replay (from the
    third epoch on; the training and the evaluation graph interleave) and without — same losses, accuracies and final
    parameters.  (Long enough for a replay that goes wrong from its SECOND launch on to show: a captured
    hipMemsetAsync does on ROCm 7.2, csrc/graph_fix.hip.)"""
    _need_gpu()
    from gnan_amd import harness
    data = _node_task(n, F, C, dense)
    loss_fn = getattr(torch.nn, loss)()
    runs = {}
    for tag, on in (("eager", False), ("graphed", True)):
        monkeypatch.setattr(harness, "GRAPHED_STEPS", on)
        m = _model(F, C)
        opt = torch.optim.Adam(m.parameters(), lr=3e-3, weight_decay=1e-4)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=7, gamma=0.5)       # the learning rate moves between replays
        hist = []
        for epoch in range(24):
            tr = harness.train_epoch(m, [data], loss_fn, opt, DEV, classify=True, compute_auc=False, is_graph_task=False)
            te = harness.test_epoch(m, [data], loss_fn, DEV, classify=True, compute_auc=(loss == "BCEWithLogitsLoss"),
                                    val_mask=True, is_graph_task=False)
            sched.step()
            hist.append(list(tr) + list(te))
        runs[tag] = (np.array(hist, dtype=np.float64), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
        if on:
            recs = [r for r in harness._steps_of(m).node.entries.values()]
            assert all(r.value["step"] is not None and r.value["step"].graph.replays >= 20 for r in recs), "nothing was replayed"
    a, b = runs["eager"], runs["graphed"]
    assert a[0].shape == b[0].shape
    assert np.allclose(a[0], b[0], rtol=2e-5, atol=1e-6), np.abs(a[0] - b[0]).max()       # two routes of the same 24 steps
    scale = max(float(v.abs().max()) for v in a[1].values())
    for k in a[1]:
        assert float((a[1][k] - b[1][k]).abs().max()) <= 2e-5 * scale, k


@pytest.mark.parametrize("n,F,C,dense,opt_cls", [(3000, 129, 1, False, "Adam"), (300, 9, 4, True, "AdamW")])
def test_captured_steps_update_the_flat_buffers_bit_for_bit(n, F, C, dense, opt_cls, monkeypatch):
    """The captured step's update is ONE fused Adam launch over the FlatMLPStore buffers (graphed.FlatAdamStep) instead of
    the optimizer's ~F x L / 70 launches: parameters AND optimizer state after twelve epochs are bit-identical to the same
    run with the optimizer's own step captured; state_dict / load_state_dict keep working, and a loaded state is noticed."""
    _need_gpu()
    from gnan_amd import graphed, harness
    data = _node_task(n, F, C, dense)
    loss_fn = torch.nn.BCEWithLogitsLoss() if C == 1 else torch.nn.CrossEntropyLoss()
    monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
    runs = {}
    for flat in (False, True):
        monkeypatch.setattr(graphed, "FLAT_OPTIMIZER_STEP", flat)
        m = _model(F, C)
        opt = getattr(torch.optim, opt_cls)(m.parameters(), lr=3e-3, weight_decay=1e-2)
        for _ in range(12):
            harness.train_epoch(m, [data], loss_fn, opt, DEV, classify=True, is_graph_task=False)
        rec = [r.value for r in harness._steps_of(m).node.entries.values() if r.value["optimizer"] is not None][0]
        assert rec["step"] is not None and rec["step"].graph.replays >= 9
        assert (rec["step"].flat is not None) == flat
        if flat:
            assert rec["step"].flat.buffers <= 12 < len(list(m.named_parameters())) and len(rec["step"].flat.P) <= 36
        runs[flat] = ({k: v.detach().clone() for k, v in m.state_dict().items()}, copy.deepcopy(opt.state_dict()), m, opt, rec)
    for k, v in runs[False][0].items():
        assert torch.equal(v, runs[True][0][k]), k
    sa, sb = runs[False][1]["state"], runs[True][1]["state"]
    assert sa.keys() == sb.keys()
    for k in sa:
        for f in ("step", "exp_avg", "exp_avg_sq"):
            assert sa[k][f].shape == sb[k][f].shape and torch.equal(sa[k][f], sb[k][f]), (k, f)
    # a state loaded into the optimizer replaces the flat views: the captured step is dropped, the next epoch runs eagerly
    # on the loaded state and is captured again
    _, _, m, opt, rec = runs[True]
    step = rec["step"]
    opt.load_state_dict(copy.deepcopy(runs[False][1]))
    harness.train_epoch(m, [data], loss_fn, opt, DEV, classify=True, is_graph_task=False)
    assert rec["step"] is None or rec["step"] is not step
    harness.train_epoch(m, [data], loss_fn, opt, DEV, classify=True, is_graph_task=False)
    assert rec["step"] is not None and rec["step"] is not step and rec["step"].flat is not None


def test_a_step_whose_tables_outgrew_the_capture_is_not_replayed(monkeypatch):
    """The unguarded route (an optimizer whose update cannot be skipped on the device, graphed.GUARDED_REPLAY = False): the tables of
    the current weights are built and compared BEFORE every replay."""
    _need_gpu()
    from gnan_amd import graphed, harness, pwl
    monkeypatch.setattr(graphed, "GUARDED_REPLAY", False)
    data = _node_task(3000, 129, 1, False)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
    m = _model(129, 1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for _ in range(4):
        harness.train_epoch(m, [data], loss_fn, opt, DEV, classify=True, is_graph_task=False)
    rec = [r.value for r in harness._steps_of(m).node.entries.values() if r.value["optimizer"] is not None][0]
    first = rec["step"]
    assert first is not None and first.graph.replays == 2
    first_graph = first.graph                                  # (a dropped step releases its graph)
    real = pwl.covers
    calls = {"n": 0}

    def covers_once_false(spec, exact):
        calls["n"] += 1
        return False if calls["n"] == 1 else real(spec, exact)
    monkeypatch.setattr(pwl, "covers", covers_once_false)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    harness.train_epoch(m, [data], loss_fn, opt, DEV, classify=True, is_graph_task=False)      # eager: the capture is dropped
    assert rec["step"] is None and first.graph is None and first_graph.replays == 2
    assert any(not torch.equal(before[k], v) for k, v in m.state_dict().items())                # ... and the step still happened
    harness.train_epoch(m, [data], loss_fn, opt, DEV, classify=True, is_graph_task=False)      # captured anew and replayed
    assert rec["step"] is not None and rec["step"] is not first and rec["step"].graph.replays == 1


@pytest.mark.parametrize("training", [True, False])
def test_tables_that_outgrow_a_captured_step_trip_its_guard(training, monkeypatch):
    """Guarded replay: the captured look-up checks ITS OWN tables on the device (gnan_pwl_check_fit) instead of the host
    building them before every replay.  Weights whose tables really outgrow the captured sizes (most hidden units switched
    off at capture time, back on afterwards): the replay runs, the guard trips, the captured update is skipped by the kernel
    (found_inf) and the step counters are put back, the epoch runs eagerly — loss, parameters and optimizer state equal a
    twin that never used a graph."""
    _need_gpu()
    from gnan_amd import graphed, harness
    from gnan_amd import functional
    monkeypatch.setattr(functional, "PWL_MIN_WORK", 1 << 18)     # (inference tabulates from 2^24 look-ups on: keep the test small)
    monkeypatch.setattr(functional, "PWL_MIN_NODES", 1 << 12)
    data = _node_task(5000, 64, 1, False)                        # 320k look-ups: the table route
    loss_fn = torch.nn.BCEWithLogitsLoss()
    monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
    assert graphed.GUARDED_REPLAY
    m = _model(64, 1)
    full = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():                                        # three of four hidden units off: few kinks, small tables
        for k, p in m.named_parameters():
            if k.startswith("fs") and (".0." in k or ".2." in k):
                p[8:] = 0.0
    opt = torch.optim.Adam(m.parameters(), lr=1e-4) if training else None

    def epoch(model, optimizer):
        if training:
            return harness.train_epoch(model, [data], loss_fn, optimizer, DEV, classify=True, is_graph_task=False)
        return harness.test_epoch(model, [data], loss_fn, DEV, classify=True, is_graph_task=False)
    for _ in range(4):
        epoch(m, opt)
    rec = [r.value for r in harness._steps_of(m).node.entries.values()][0]
    step = rec["step"]
    assert step is not None and step.graph.replays == 2 and step.guard is not None and step.graph.builds
    fits_calls = {"n": 0}
    real_fits = graphed.GraphedCallable.fits
    monkeypatch.setattr(graphed.GraphedCallable, "fits", lambda self: (fits_calls.__setitem__("n", fits_calls["n"] + 1), real_fits(self))[1])
    epoch(m, opt)
    assert step.graph.replays == 3 and fits_calls["n"] == 0      # no table build before the replay any more
    with torch.no_grad():                                        # all hidden units back: ~4x the pieces per feature
        for k, p in m.named_parameters():
            p.copy_(full[k])
    twin = copy.deepcopy(m)
    harness.release_steps(twin)
    topt = None
    if training:
        topt = torch.optim.Adam(twin.parameters(), lr=1e-4, capturable=True, fused=True)
        topt.load_state_dict(copy.deepcopy(opt.state_dict()))
    monkeypatch.setattr(harness, "GRAPHED_STEPS", False)
    want = epoch(twin, topt)
    monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
    graph = step.graph
    got = epoch(m, opt)
    assert graph.replays == 4 and rec["step"] is None            # replayed, guard tripped, dropped; the epoch ran eagerly
    assert abs(got[0] - want[0]) <= 1e-6 * abs(want[0]) and got[1] == want[1]
    for (k, a), (_, b) in zip(m.state_dict().items(), twin.state_dict().items()):
        assert torch.equal(a, b), k
    if training:
        sa, sb = opt.state_dict()["state"], topt.state_dict()["state"]
        for k in sa:
            assert float(sa[k]["step"]) == float(sb[k]["step"]) and torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]), k
    for _ in range(2):
        epoch(m, opt)                                            # captured again with the larger tables
    assert rec["step"] is not None and rec["step"] is not step


def test_moved_parameters_invalidate_the_capture(monkeypatch):
    _need_gpu()
    from gnan_amd import harness
    data = _node_task(3000, 129, 1, False)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
    m = _model(129, 1)
    for _ in range(4):
        out = harness.test_epoch(m, [data], loss_fn, DEV, classify=True, is_graph_task=False)
    rec = [r.value for r in harness._steps_of(m).node.entries.values()][0]
    assert rec["step"] is not None and rec["step"].graph.replays == 2
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.mul_(1.5)                                   # in place: the graph reads the new values
    changed = harness.test_epoch(m, [data], loss_fn, DEV, classify=True, is_graph_task=False)
    assert rec["step"].graph.replays == 3 and abs(changed[0] - out[0]) > 1e-6
    m.double().float()                                    # re-homes every parameter: new storage
    again = harness.test_epoch(m, [data], loss_fn, DEV, classify=True, is_graph_task=False)
    assert rec["step"] is None and abs(again[0] - changed[0]) <= 1e-5 * max(1.0, abs(changed[0]))


def test_captured_memsets_are_swapped_for_kernels():
    """A hipMemsetAsync captured into a hipGraph fills with garbage from the second replay on (ROCm 7.2);
    gnan_graph_replace_memsets swaps the node for a kernel node and the graph replays correctly ever after."""
    _need_gpu()
    import ctypes
    from gnan_amd import _lib
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    for offset, nbytes in [(16, 8), (0, 1024), (64, 64), (256, 4096)]:
        buf = torch.full(((offset + nbytes) // 4 + 16,), 7, dtype=torch.int32, device=DEV)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            assert hip.hipMemsetAsync(buf.data_ptr() + offset, 0, nbytes, torch.cuda.current_stream().cuda_stream) == 0
            buf.add_(1)
        k = ctypes.c_int32(0)
        _lib.check(_lib.lib().gnan_graph_replace_memsets(g.raw_cuda_graph(), ctypes.byref(k)), "gnan_graph_replace_memsets")
        assert k.value == 1
        g.instantiate()
        lo, hi = offset // 4, (offset + nbytes) // 4
        for r in range(4):
            g.replay()
            torch.cuda.synchronize()
            assert bool((buf[lo:hi] == 1).all()), (offset, nbytes, r, buf[lo:hi][:4].tolist())
            assert int(buf[-1]) == 8 + r and (lo == 0 or int(buf[lo - 1]) == 8 + r)      # the neighbours are not touched


def _graph_task(n_graphs, F, sizes, seed=0):
    from oracle import gnan_oracle as O
    rng = np.random.default_rng(seed)
    data = []
    for i in range(n_graphs):
        n = int(sizes[i % len(sizes)])
        tree = np.stack([np.arange(1, n), rng.integers(0, np.arange(1, n))])              # a random tree: connected
        extra = np.stack([rng.integers(0, n, n // 4 + 1), rng.integers(0, n, n // 4 + 1)])
        ei = np.concatenate([tree, extra], axis=1)
        if i % 5 == 0 and n > 6:
            ei = ei[:, ei.max(0) < n - 2]                                                 # two isolated nodes: a rest shell
        nd, norm = O.pre_process_dense(np.concatenate([ei, ei[::-1]], axis=1), n)
        x = torch.zeros(n, F)
        x[torch.arange(n), torch.from_numpy(rng.integers(0, F - 1, n))] = 1.0             # one-hot atom types + ones column
        x[:, -1] = 1.0
        y = torch.tensor([[1.0 if rng.random() < 0.5 else -1.0]])                         # {-1, +1} targets (trainer.py:35-38)
        data.append(Bag(x=x.to(DEV), y=y.to(DEV), edge_index=None, node_distances=nd.to(DEV), normalization_matrix=norm.to(DEV)))
    return data


@pytest.mark.parametrize("readout", [0, 2])
def test_graph_task_steps_replayed_per_shape_match_eager_steps(readout, monkeypatch):
    """Graph-level task, batch_size = 1: one captured step per graph shape, every later graph of the shape copied into its
    static buffers and replayed.  Two copies of the model walk the same graphs in lock-step — one eagerly, one through the
    replayer — and after EVERY step losses, parameters and optimizer moments must agree; the replayed copy is then reset to
    the eager one's state, so every replayed step is checked from the same starting point (with the NAM read-out and O(1)
    random weights the trajectory itself is chaotic: for-each and fused Adam — no graphs involved — drift apart by 20 %
    of the epoch loss within three epochs, which says nothing about a step being right)."""
    _need_gpu()
    from gnan_amd import harness
    from gnan_amd.models import TensorGNAN
    monkeypatch.setattr(harness, "SLOT_STEPS", False)          # (the per-shape steps: what graphs beyond the slots and the NAM read-out take)
    F = 15
    graphs = _graph_task(45, F, sizes=[12, 30, 12, 23, 30, 12, 41])
    loss_fn = torch.nn.BCEWithLogitsLoss()

    def make():
        torch.manual_seed(0)
        m = TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=readout, device=DEV)
        with torch.no_grad():
            for _, p in m.named_parameters():
                p.copy_(torch.randn(p.shape) * 0.5)
        m = m.to(DEV).eval()
        return m, torch.optim.Adam(m.parameters(), lr=2e-3, capturable=True, fused=True)
    (ma, oa), (mb_, ob) = make(), make()
    replayed = worst = 0
    for epoch in range(3):
        for g in graphs:
            monkeypatch.setattr(harness, "GRAPHED_STEPS", False)
            ra = harness.train_epoch(ma, [g], loss_fn, oa, DEV, classify=True, is_graph_task=True)
            monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
            rb = harness.train_epoch(mb_, [g], loss_fn, ob, DEV, classify=True, is_graph_task=True)
            assert abs(ra[0] - rb[0]) <= 1e-5 * max(1.0, abs(ra[0])) and ra[1] == rb[1], (epoch, ra, rb)
            with torch.no_grad():
                for pa, pb in zip(ma.parameters(), mb_.parameters()):
                    scale = float(pa.abs().max()) + 1e-12
                    worst = max(worst, float((pa - pb).abs().max()) / scale)
                    pb.copy_(pa)
                    for key in ("exp_avg", "exp_avg_sq", "step"):
                        ob.state[pb][key].copy_(oa.state[pa][key])
            assert worst <= 1e-5, (epoch, worst)
    steps = harness._steps_of(mb_).graph
    replayed = sum(r["step"].step.graph.replays for r in steps.buckets.values() if r["step"] is not None)
    assert replayed >= 70, replayed                                # 135 steps, a dozen shapes, two eager sightings each


@pytest.mark.parametrize("task", ["graph", "node"])
def test_optimizers_the_flat_update_declines_are_replayed_with_their_own_step(task, monkeypatch):
    """An optimizer over a SUBSET of the parameters (rho frozen) or with two parameter groups is not re-homed
    (graphed.FlatAdamStep declines): its own fused step is captured, such a step is checked BEFORE every replay instead of
    guarded on the device, the small-graph backward leaves the frozen MLP alone — and the replayed epochs still equal the eager
    ones, parameter for parameter."""
    _need_gpu()
    from gnan_amd import harness
    from gnan_amd.models import TensorGNAN
    monkeypatch.setattr(harness, "SLOT_STEPS", False)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    if task == "graph":
        F = 15
        loader = _graph_task(30, F, sizes=[12, 30, 12, 23, 30])
    else:
        F = 129
        loader = [_node_task(3000, F, 1, False)]

    def make(frozen):
        torch.manual_seed(0)
        m = TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=(task == "graph"), readout_n_layers=0, device=DEV)
        with torch.no_grad():
            for _, p in m.named_parameters():
                p.copy_(torch.randn(p.shape) * 0.3)
        m = m.to(DEV).eval()
        if frozen:
            for _, p in m.rho.named_parameters():
                p.requires_grad_(False)
            opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, capturable=True, fused=True)
        else:
            fs, rho = list(m.fs.parameters()), list(m.rho.parameters())
            opt = torch.optim.Adam([{"params": fs}, {"params": rho, "lr": 3e-4}], lr=1e-3, capturable=True, fused=True)
        return m, opt
    for frozen in (True, False):
        runs = {}
        for on in (False, True):
            monkeypatch.setattr(harness, "GRAPHED_STEPS", on)
            m, opt = make(frozen)
            hist = [harness.train_epoch(m, loader, loss_fn, opt, DEV, classify=True, is_graph_task=(task == "graph"))[0]
                    for _ in range(6)]
            runs[on] = (hist, {k: v.detach().clone() for k, v in m.state_dict().items()}, m)
        assert np.allclose(runs[False][0], runs[True][0], rtol=1e-5, atol=1e-7), (frozen, runs[False][0], runs[True][0])
        scale = max(float(v.abs().max()) for v in runs[False][1].values())
        for k, v in runs[False][1].items():
            assert float((v - runs[True][1][k]).abs().max()) <= 1e-5 * scale, (frozen, k)
        store = harness._steps_of(runs[True][2])
        steps = ([r["step"].step for r in store.graph.buckets.values() if r["step"] is not None] if task == "graph"
                 else [r.value["step"] for r in store.node.entries.values() if r.value["step"] is not None])
        assert steps and all(s.flat is None and s.guard is None and s.graph.replays >= 1 for s in steps), (frozen, len(steps))
        if frozen:
            assert all(p.grad is None for p in runs[True][2].rho.parameters())


def test_captured_steps_die_with_their_model_and_hand_the_optimizer_back(monkeypatch):
    """Cross-validation loops build a fresh model per fold and seed (main.py): every run's captured graphs (and their
    private memory pools) must go when the model goes — device memory returns to the baseline — and an optimizer that was
    switched to its capturable mode gets its own settings (a float learning rate, its foreach / fused choice) back when the
    steps are released."""
    _need_gpu()
    import gc
    from gnan_amd import harness
    monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
    data = _node_task(3000, 129, 1, False)
    loss_fn = torch.nn.BCEWithLogitsLoss()

    def one_fold(release):
        m = _model(129, 1)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        for _ in range(5):
            harness.train_epoch(m, [data], loss_fn, opt, DEV, classify=True, is_graph_task=False)
            harness.test_epoch(m, [data], loss_fn, DEV, classify=True, val_mask=True, is_graph_task=False)
        recs = [r.value for r in harness._steps_of(m).node.entries.values()]
        assert len(recs) == 2 and all(r["step"] is not None and r["step"].graph.replays >= 2 for r in recs)
        assert torch.is_tensor(opt.param_groups[0]["lr"]) and opt.param_groups[0]["capturable"]
        if release:
            harness.release_steps(m)
            g = opt.param_groups[0]
            assert isinstance(g["lr"], float) and abs(g["lr"] - 1e-3) < 1e-12 and not g["capturable"] and not g["fused"]
            harness.train_epoch(m, [data], loss_fn, opt, DEV, classify=True, is_graph_task=False)   # and it still steps
            harness.release_steps(m)
        return opt

    survivor = one_fold(False)                       # the optimizer outlives its model
    gc.collect()
    g0 = survivor.param_groups[0]
    assert isinstance(g0["lr"], float) and not g0["capturable"], "the collected model's steps did not hand the optimizer back"
    del survivor
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    reserved = torch.cuda.memory_reserved()
    for fold in range(4):
        one_fold(fold % 2 == 1)
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        assert torch.cuda.memory_allocated() <= base + (1 << 20), (fold, torch.cuda.memory_allocated(), base)
        assert torch.cuda.memory_reserved() <= reserved + (8 << 20), (fold, torch.cuda.memory_reserved(), reserved)


def test_an_edited_adjacency_is_not_replayed(monkeypatch):
    """The step record is keyed on every tensor the step reads: swapping the graph on the same Data object runs the new
    graph (eagerly, then captured anew), never the captured old one."""
    _need_gpu()
    from gnan_amd import harness
    from gnan_amd import synthetic as syn
    monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
    n = 3000
    data = _node_task(n, 129, 1, False)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    m = _model(129, 1)
    for _ in range(4):
        first = harness.test_epoch(m, [data], loss_fn, DEV, classify=True, is_graph_task=False)
    src, dst = syn.uniform_edges(n, 2 * n, 99, DEV)
    other = syn.hop1_csr(src, dst, n)
    data.gnan_graph = other
    monkeypatch.setattr(harness, "GRAPHED_STEPS", False)
    want = harness.test_epoch(m, [data], loss_fn, DEV, classify=True, is_graph_task=False)
    monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
    for _ in range(4):
        got = harness.test_epoch(m, [data], loss_fn, DEV, classify=True, is_graph_task=False)
        assert abs(got[0] - want[0]) <= 1e-6 * max(1.0, abs(want[0])) and abs(got[0] - first[0]) > 1e-6


@pytest.mark.parametrize("readout", [0, 2])
def test_graph_task_evaluation_passes_are_replayed_per_shape(readout, monkeypatch):
    """test_epoch on a graph-level task (trainer.py:89-154 runs after every training epoch): from the third graph of a shape on,
    the forward + loss + hit count of a graph is one replayed hipGraph; same epoch returns as the eager pass."""
    _need_gpu()
    from gnan_amd import harness
    from gnan_amd.models import TensorGNAN
    monkeypatch.setattr(harness, "SLOT_STEPS", False)
    F = 15
    graphs = _graph_task(60, F, sizes=[12, 30, 12, 23, 30, 12, 41])
    loss_fn = torch.nn.BCEWithLogitsLoss()
    torch.manual_seed(0)
    m = TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=readout, device=DEV)
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    m = m.to(DEV).eval()
    monkeypatch.setattr(harness, "GRAPHED_STEPS", False)
    want = harness.test_epoch(m, graphs, loss_fn, DEV, classify=True, val_mask=True, is_graph_task=True)
    monkeypatch.setattr(harness, "GRAPHED_STEPS", True)
    for epoch in range(3):
        got = harness.test_epoch(m, graphs, loss_fn, DEV, classify=True, val_mask=True, is_graph_task=True)
        assert abs(got[0] - want[0]) <= 1e-6 * max(1.0, abs(want[0])) and got[1] == want[1], (epoch, got, want)
    steps = harness._steps_of(m).graph_eval
    replays = sum(r["step"].step.graph.replays for r in steps.buckets.values() if r["step"] is not None)
    assert replays >= 120, replays                             # 180 graphs, a handful of shapes, two eager sightings each
    with torch.no_grad():                                      # the weights move: the replay reads the new values
        for _, p in m.named_parameters():
            p.mul_(1.1)
    moved = harness.test_epoch(m, graphs, loss_fn, DEV, classify=True, val_mask=True, is_graph_task=True)
    monkeypatch.setattr(harness, "GRAPHED_STEPS", False)
    moved_eager = harness.test_epoch(m, graphs, loss_fn, DEV, classify=True, val_mask=True, is_graph_task=True)
    assert abs(moved[0] - moved_eager[0]) <= 1e-6 * max(1.0, abs(moved_eager[0])) and abs(moved[0] - want[0]) > 1e-6
    harness.release_steps(m)


# ---------------------------------------------------------------------------------------------------------- graph slots
@pytest.mark.parametrize("normalize", [True, False])
def test_slot_graph_forward_backward_equal_the_one_launch_path(normalize):
    """small_graph.SlotGraph: one graph in device slots (its size known to the device only), forwarded and back-propagated by
    the batched kernels with the shell sizes (ABI 42) == the per-graph one-launch kernels on the same graph: graphs of 1 to
    128 nodes, with unreachable pairs, one and three output channels; and == the float64 oracle."""
    _need_gpu()
    from gnan_amd.models import TensorGNAN
    from gnan_amd.small_graph import SlotGraph, slot_graph_applies, slot_graph_forward
    from oracle import gnan_oracle as O
    F = 7
    graphs = _graph_task(12, F, sizes=[1, 5, 128, 64, 65, 30, 17, 100, 2, 41, 12, 90], seed=3)
    for C in (1, 3):
        torch.manual_seed(C)
        m = TensorGNAN(F, C, 3, hidden_channels=16, is_graph_task=True, readout_n_layers=0, normalize_rho=normalize, device=DEV)
        with torch.no_grad():
            for _, p in m.named_parameters():
                p.copy_(torch.randn(p.shape) * 0.5)
        m = m.to(DEV).eval()
        slot = SlotGraph(F, DEV, use_cnt=normalize)
        for d in graphs:
            g = m.hop_graph(d)
            assert slot.fits(g, d.x)
            m.zero_grad()
            y0 = m.forward(d)
            y0.pow(2).sum().backward()
            g0 = {k: p.grad.clone() for k, p in m.named_parameters()}
            slot.load(g, d.x)
            m.zero_grad()
            f, rho = m._stacked("fs", m.fs), m._stacked("rho", [m.rho])
            assert slot_graph_applies(slot, f, rho)
            y1 = slot_graph_forward(slot, f, rho, normalize)
            y1.pow(2).sum().backward()
            assert y1.shape == y0.shape == (C, 1)
            scale = float(y0.abs().max()) + 1e-30
            assert float((y1 - y0).abs().max()) <= 2e-6 * scale, (C, d.x.shape, float((y1 - y0).abs().max()) / scale)
            gscale = max(float(v.abs().max()) for v in g0.values()) + 1e-30
            for k, p in m.named_parameters():
                assert float((p.grad - g0[k]).abs().max()) <= 2e-6 * gscale, (C, d.x.shape, k)
            sd64 = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
            truth = O.tensor_gnan_forward_models(d.x.cpu().double(), d.node_distances.cpu().double(),
                                                 d.normalization_matrix.cpu().double(), sd64, normalize, True, 0)
            assert O.rel_err(y1.detach().cpu(), truth) <= 1e-5


def test_graph_task_epochs_through_one_slot_step(monkeypatch):
    """Graph-level task, batch_size = 1 (trainer.py:23-86): with graph slots the harness captures ONE training step (and one
    evaluation step) per model, after two eager steps of the FIRST epoch, and replays it for every graph of up to 128 nodes
    whatever its shape; larger graphs keep their own routes.  Epoch returns and parameters == the eager loop's."""
    _need_gpu()
    from gnan_amd import harness
    from gnan_amd.models import TensorGNAN
    F = 15
    graphs = _graph_task(40, F, sizes=[12, 30, 9, 23, 130, 12, 41, 77, 5, 128])
    loss_fn = torch.nn.BCEWithLogitsLoss()

    def make():
        torch.manual_seed(0)
        m = TensorGNAN(F, 1, 3, hidden_channels=32, is_graph_task=True, readout_n_layers=0, device=DEV)
        with torch.no_grad():
            for _, p in m.named_parameters():
                p.copy_(torch.randn(p.shape) * 0.3)
        m = m.to(DEV).eval()
        return m, torch.optim.Adam(m.parameters(), lr=1e-3)
    runs = {}
    for on in (False, True):
        monkeypatch.setattr(harness, "GRAPHED_STEPS", on)
        m, opt = make()
        hist = []
        for epoch in range(3):
            tr = harness.train_epoch(m, graphs, loss_fn, opt, DEV, classify=True, is_graph_task=True)
            te = harness.test_epoch(m, graphs, loss_fn, DEV, classify=True, val_mask=True, is_graph_task=True)
            hist.append((tr[0], tr[1], te[0], te[1]))
        runs[on] = (hist, {k: v.detach().clone() for k, v in m.state_dict().items()}, m)
    for a, b in zip(runs[False][0], runs[True][0]):
        assert abs(a[0] - b[0]) <= 1e-5 * max(1.0, abs(a[0])) and a[1] == b[1], (a, b)
        assert abs(a[2] - b[2]) <= 1e-5 * max(1.0, abs(a[2])) and a[3] == b[3], (a, b)
    scale = max(float(v.abs().max()) for v in runs[False][1].values())
    for k, v in runs[False][1].items():
        assert float((v - runs[True][1][k]).abs().max()) <= 1e-5 * scale, k
    store = harness._steps_of(runs[True][2])
    for steps, epochs in ((store.graph, 3), (store.graph_eval, 3)):
        assert 2 <= len(steps.slots) <= 6 and {t[0] for t in steps.slots} == {64, 128}      # a step per (node, hop-code) tier, whatever the shapes
        n_fit = sum(d.x.shape[0] <= 128 for d in graphs)
        replays = sum(s.step.graph.replays for s in steps.slots.values())
        assert replays >= epochs * n_fit - 3, (replays, n_fit)         # from the third step of the first epoch on
        assert all(s.step.graph.kernel_nodes <= 7 for s in steps.slots.values())
        # the 130-node graphs: their own per-shape steps (or eager), never the slots; no per-shape capture for the others
        assert all(key[0] > 128 for key, rec in steps.buckets.items() if rec["step"] is not None)
    harness.release_steps(runs[True][2])
