"""f-3 on the GPU: harness.train_epoch / test_epoch driving the gnan_amd modules against numbers captured from the
REFERENCE's trainer.train_epoch / test_epoch driving the REFERENCE's GNAN classes (tests/golden/make_golden.py, cases
310-317: inputs, initial state_dict, per-epoch returns, updated state_dict) — SGD steps and Adam runs long enough for the
harness to capture the step into a hipGraph and replay it; and f-4: the interpretability exports from the kernels' tables."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, golden_names
from oracle import gnan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    batches = []
    for b in range(meta["n_batches"]):
        batches.append(Bag(**{k.split("/", 1)[1]: torch.from_numpy(z[k]).to(DEV) for k in z.files if k.startswith(f"b{b}/")}))
    return z, meta, batches


def _module(meta):
    from gnan_amd import GNAN as standalone
    from gnan_amd import models
    kw = dict(in_channels=meta["F"], out_channels=meta["C"], hidden_channels=meta["H"], bias=True, dropout=0.0, device=DEV)
    v = meta["model"]
    if v.startswith("standalone_tensor"):
        return standalone.TensorGNAN(n_layers=meta["L"], normalize_rho=True, is_graph_task=v.endswith("graph"), **kw)
    if v.startswith("models_tensor"):
        return models.TensorGNAN(n_layers=meta["L"], normalize_rho=True, is_graph_task=v.endswith("graph"),
                                 rho_per_feature=meta["rho_per_feature"], readout_n_layers=0, **kw)
    if v == "models_gnan":
        return models.GNAN(num_layers=meta["L"], normalize_rho=True, rho_per_feature=meta["rho_per_feature"], **kw)
    raise ValueError(v)


@pytest.mark.parametrize("graphed", [False, True])
@pytest.mark.parametrize("name", golden_names("trainer_gnan"))
def test_harness_epochs_on_gnan_modules_match_the_reference_trainer(name, graphed, monkeypatch):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from gnan_amd import harness
    z, meta, batches = _load(name)
    if graphed and meta["optimizer"] == "SGD":
        pytest.skip("torch's SGD has no capturable mode: the harness keeps the eager loop (covered by graphed=False)")
    monkeypatch.setattr(harness, "GRAPHED_STEPS", graphed)
    model = _module(meta)
    model.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0/")}, strict=True)
    model = model.to(DEV).train()
    loss_fn = getattr(torch.nn, meta["loss"])()
    opt = (torch.optim.SGD if meta["optimizer"] == "SGD" else torch.optim.Adam)(model.parameters(), lr=meta["lr"])
    epochs = meta["epochs"]
    hist_tr = z["train_hist"] if epochs > 1 else z["train_ret"][None]
    hist_te = z["test_hist"] if epochs > 1 else z["test_ret"][None]
    for e in range(epochs):
        tr = harness.train_epoch(model, batches, loss_fn, opt, DEV, classify=meta["classify"], compute_auc=False,
                                 is_graph_task=meta["graph"])
        te = harness.test_epoch(model, batches, loss_fn, DEV, classify=meta["classify"], compute_auc=False, val_mask=True,
                                is_graph_task=meta["graph"])
        # every trainer fixture carries the reference's float64 twin run (tests/golden/make_golden_run.py): the rule of SURVEY
        # section 8c on the trajectory — |loss - loss64| <= max(1e-5, the float32 reference's own gap) of the loss scale;
        # accuracies are hit counts
        for got, h32, h64 in ((tr, hist_tr, z["train_hist64"]), (te, hist_te, z["test_hist64"])):
            scale64 = np.abs(h64[:, 0]).max()
            ref_gap = np.abs(h32[:, 0] - h64[:, 0]).max() / scale64
            err = abs(float(got[0]) - h64[e, 0]) / scale64
            assert err <= max(1e-5, ref_gap), (e, err, ref_gap)
            assert abs(float(got[1]) - h32[e, 1]) <= 1e-6, (e, got, h32[e])        # (a float32 quotient in the fixture)
    assert not model.training                                       # trainer.py:97 leaves eval mode on
    scale = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("sd1/"))
    # parameters after the run: against the float64 run, bounded by the float32 run's own gap
    gap = max(float(np.abs(z["sd1/" + k] - z["sd1_64/" + k]).max()) for k in model.state_dict()) / scale
    for k, v in model.state_dict().items():
        assert float(np.abs(v.cpu().numpy() - z["sd1_64/" + k]).max()) <= max(1e-5, gap) * scale, (k, gap)
    if graphed:
        store = harness._steps_of(model)
        if meta["graph"]:
            replays = sum(r["step"].step.graph.replays for r in store.graph.buckets.values() if r["step"] is not None)
            replays += sum(sl.step.graph.replays for sl in store.graph.slots.values())     # graph slots (graphed.SlotGraphStep)
        else:
            replays = sum(r.value["step"].graph.replays for r in store.node.entries.values() if r.value["step"] is not None)
        assert replays >= 3, "the captured step was never replayed"
        harness.release_steps(model)


def test_interpretability_exports_come_from_the_kernels_tables():
    """f-4 (notebook cells 4-9): rho on the distinct distances, the shape functions on a value grid and the
    f_k(1) * rho(d) heat-map — exported from the piecewise-linear tables the kernels look up (gnan_pwl_build +
    gnan_fpwl_fwd on the grid), compared with the float64 oracle."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from gnan_amd import interpret
    from gnan_amd.models import TensorGNAN
    torch.manual_seed(0)
    F, C = 15, 3
    m = TensorGNAN(F, C, 3, hidden_channels=64, rho_per_feature=True, device=DEV)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else (2.0 / sum(p.shape)) ** 0.5))
        m.fs[2][0].bias.zero_()                                     # a feature whose kinks all sit at 0 (the grid contains 0)
    m = m.to(DEV).eval()
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    r = interpret.rho_curve(m, 6)
    assert r.shape == (8, C) and r.is_cuda
    assert O.rel_err(r.cpu(), O.rho_lut(sd, 8, dtype=torch.float64)) <= 1e-5
    grid = torch.linspace(-2, 3, 501)
    f = interpret.shape_functions(m, grid)                          # [F, 501, C]
    want = O.feature_mlps(grid.double().view(-1, 1).expand(-1, F).contiguous(), sd).permute(1, 0, 2)
    assert f.shape == (F, 501, C) and O.rel_err(f.cpu(), want) <= 1e-5
    exported = interpret.shape_function_tables(m)                   # the tables themselves: exact breakpoints, values, slopes
    assert exported.off.numel() == F + 1 and exported.val.shape[1] == C
    h = interpret.contribution_heatmap(m, 6)
    f1 = O.feature_mlps(torch.ones(1, F, dtype=torch.float64), sd)[0]              # [F, C]
    assert h.shape == (F, 7, C)
    assert O.rel_err(h.cpu(), f1.unsqueeze(1) * O.rho_lut(sd, 8, dtype=torch.float64)[:7].unsqueeze(0)) <= 1e-5


@pytest.mark.parametrize("kind,n_rows,C,masked", [("bce", 1, 1, False), ("bce", 37, 1, True), ("bce", 300_000, 1, True),
                                                   ("ce", 50, 4, False), ("ce", 5000, 7, True), ("ce", 200_000, 3, True),
                                                   ("ce", 3, 40, False), ("ce", 100_000, 40, True), ("ce", 5001, 128, True),
                                                   ("ce", 777, 9, True), ("ce", 64, 130, False)])
def test_fused_loss_step_matches_torch(kind, n_rows, C, masked):
    """gnan_loss_step (row selection, mean loss, gradient w.r.t. the logits, hit count, running totals: one launch, two past
    2048 rows) == nn.BCEWithLogitsLoss / nn.CrossEntropyLoss + the trainer's accuracy rule (trainer.py:5-20, 61-71) on the
    same rows; strided logits, float64 truth for the loss."""
    _need_gpu()
    from gnan_amd import _lib
    from gnan_amd.losses import loss_kind, loss_step
    g = torch.Generator().manual_seed(n_rows + C)
    wide = (torch.randn(n_rows, C + 2, generator=g) * 3).to(DEV)
    logits = wide[:, 1:1 + C].clone().requires_grad_(True)                      # contiguous leaf ...
    strided = wide[:, 1:1 + C].detach().requires_grad_(True)                    # ... and a strided one
    idx = torch.nonzero(torch.rand(n_rows, generator=g) < 0.6).flatten().to(DEV) if masked else None
    n = n_rows if idx is None else int(idx.numel())
    if kind == "bce":
        loss_fn, labels = torch.nn.BCEWithLogitsLoss(), (torch.rand(n, generator=g) < 0.5).float().to(DEV)
    else:
        loss_fn, labels = torch.nn.CrossEntropyLoss(), torch.randint(0, C, (n,), generator=g).to(DEV)
    k = loss_kind(loss_fn, logits)
    assert k == (_lib.LOSS_BCE_LOGITS if kind == "bce" else _lib.LOSS_CROSS_ENTROPY)
    total, hit_total = torch.full((), 2.5, device=DEV), torch.full((), 10.0, device=DEV)
    loss, hits = loss_step(logits, labels, k, index=idx, loss_sum=total, hits_sum=hit_total)
    (3.0 * loss).backward()
    # torch on the same rows
    ref_in = wide[:, 1:1 + C].detach().clone().requires_grad_(True)
    picked = ref_in if idx is None else ref_in.index_select(0, idx)
    want = loss_fn(picked.flatten(), labels) if kind == "bce" else loss_fn(picked, labels)
    (3.0 * want).backward()
    want_hits = int(((torch.sigmoid(picked.detach()).reshape(-1) > 0.5) == labels).sum()) if kind == "bce" \
        else int((picked.detach().argmax(-1) == labels).sum())
    p64 = picked.detach().double()
    truth = float(torch.nn.functional.binary_cross_entropy_with_logits(p64.flatten(), labels.double())) if kind == "bce" \
        else float(torch.nn.functional.cross_entropy(p64, labels))
    assert abs(float(loss.detach()) - truth) <= 2e-6 * abs(truth) and abs(float(want.detach()) - truth) <= 1e-5 * abs(truth)
    assert int(hits) == want_hits
    scale = float(ref_in.grad.abs().max())
    assert float((logits.grad - ref_in.grad).abs().max()) <= 2e-6 * scale
    assert abs(float(total) - 2.5 - float(loss.detach())) <= 1e-6 * max(1.0, abs(float(loss.detach()))) and float(hit_total) == 10.0 + want_hits
    loss2, none = loss_step(strided, labels, k, index=idx, want_hits=False)
    loss2.backward()
    assert none is None and float(loss2) == float(loss)
    assert torch.equal(strided.grad * 3.0, logits.grad) or float((strided.grad * 3.0 - logits.grad).abs().max()) <= 1e-6 * scale
    with torch.no_grad():                                                        # evaluation: no gradient buffer
        loss3, hits3 = loss_step(logits, labels, k, index=idx)
    assert float(loss3) == float(loss) and int(hits3) == want_hits


def test_fused_loss_step_declines_other_losses():
    _need_gpu()
    from gnan_amd.losses import loss_kind
    x1, x4 = torch.zeros(3, 1, device=DEV), torch.zeros(3, 4, device=DEV)
    assert loss_kind(torch.nn.MSELoss(), x1) is None
    assert loss_kind(torch.nn.BCEWithLogitsLoss(pos_weight=torch.ones(1, device=DEV)), x1) is None
    assert loss_kind(torch.nn.BCEWithLogitsLoss(reduction="sum"), x1) is None
    assert loss_kind(torch.nn.BCEWithLogitsLoss(), x4) is None
    assert loss_kind(torch.nn.CrossEntropyLoss(label_smoothing=0.1), x4) is None
    assert loss_kind(torch.nn.CrossEntropyLoss(weight=torch.ones(4, device=DEV)), x4) is None
    assert loss_kind(torch.nn.CrossEntropyLoss(), x4.cpu()) is None
    assert loss_kind(torch.nn.CrossEntropyLoss(), x4.double()) is None


# ---------------------------------------------------------------------------------------------------------- run level
def _run_loaders(z, meta):
    def group(tag, count):
        return [Bag(**{k.split("/", 1)[1]: torch.from_numpy(z[k]).to(DEV) for k in z.files if k.startswith(f"{tag}{b}/")})
                for b in range(count)]
    train = group("train", meta["n_train"])
    if meta["shared_loaders"]:
        return train, train, train
    return train, group("val", meta["n_val"]), group("test", meta["n_test"])


@pytest.mark.parametrize("name", golden_names("run_exp"))
def test_run_exp_reproduces_the_reference_run(name, tmp_path):
    """gnan_amd.run.run_exp (Adam + ReduceLROnPlateau on the training loss + early stopping + checkpoints, main.py:139-303)
    on the gnan_amd modules against the REFERENCE's run_exp driving the reference's classes (golden 320-321, float32 and
    float64).  The epochs are replayed from captured hipGraphs from the third one on; the learning rate the scheduler
    rewrites must reach the captured Adam update through the optimizer's device-tensor lr — no new capture.
    Tolerance by the rule of SURVEY section 8c: |loss - loss64| <= max(1e-5, the float32 reference's own gap) of the scale."""
    _need_gpu()
    from gnan_amd import harness, run
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    train, val, test = _run_loaders(z, meta)
    model = run.build_model(meta["F"], meta["C"], meta["L"], meta["H"], 0.0, DEV, int(meta["rho_per_feature"]), 1, meta["graph"], 0)
    model.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0/")}, strict=True)
    seen = []
    real_train = harness.train_epoch

    def spy(net, *a, **kw):
        out = real_train(net, *a, **kw)
        store = harness._steps_of(net)
        steps = [id(e.value.get("step")) for e in store.node.entries.values() if e.value.get("step") is not None]
        if store.graph is not None:
            steps += [id(r["step"]) for r in store.graph.buckets.values() if r.get("step") is not None]
            steps += [id(sl) for sl in store.graph.slots.values()]
        opt = kw.get("optimizer")
        seen.append((sorted(steps), torch.is_tensor(opt.param_groups[0]["lr"])))
        return out

    harness.train_epoch = spy
    try:
        runs = run.run_exp(train, val, test, meta["F"], [meta["seed"]], meta["L"], meta["early_stop_flag"], 0.0, "gnan",
                           meta["epochs"], 0, meta["wd"], meta["H"], meta["lr"], 1e-5, meta["data_name"], "RUN",
                           int(meta["rho_per_feature"]), 1, meta["graph"], meta["num_classes"], meta["C"],
                           patience=meta["patience"], model=model, device=DEV, checkpoint_dir=str(tmp_path), log=lambda *_: None)
    finally:
        harness.train_epoch = real_train
    r = runs[0]
    h32, h64 = z["hist32"], z["hist64"]
    n = len(r["epochs"])
    assert n == meta["epochs_run"] and r["stopped"].split(" at ")[0].startswith(meta["stopped"].split()[0])
    got = np.array([[e["train_loss"], e["train_acc"], e["val_loss"], e["val_acc"], e["test_loss"], e["test_acc"], e["lr"]]
                    for e in r["epochs"]])
    for col, what in ((0, "train loss"), (2, "val loss"), (4, "test loss")):
        scale = np.abs(h64[:n, col]).max()
        ref_gap = np.abs(h32[:n, col] - h64[:n, col]).max() / scale
        err = np.abs(got[:, col] - h64[:n, col]).max() / scale
        assert err <= max(1e-5, ref_gap), (what, err, ref_gap)
    assert np.allclose(got[:, 6], h32[:n, 6], rtol=1e-6, atol=0)      # the scheduler's learning rates (a float32 device tensor here)
    moved = lambda lr: np.nonzero(np.abs(np.diff(lr)) > 1e-6 * lr[:-1])[0]                          # (float <-> float32 tensor aside)
    assert np.array_equal(moved(got[:, 6]), moved(h32[:n, 6]))                                     # changed at the same epochs
    for col in (1, 3, 5):                                              # accuracies: hit counts over sample counts, identical
        assert np.allclose(got[:, col], h32[:n, col], rtol=1e-6, atol=0), col
    assert [(e, f) for e, f in r["checkpoints"]] == [tuple(c) for c in meta["checkpoints"]]
    # where the run ended (same keys as the reference's): against the float64 twin run's parameters, within max(1e-5, the
    # float32 reference's own distance from them) of the largest parameter — the rule of SURVEY section 8c, hundreds of epochs in
    sd = r["model"].state_dict()
    scale = max(float(np.abs(z["sd1_64/" + k]).max()) for k in sd)
    gap = max(float(np.abs(z["sd1/" + k] - z["sd1_64/" + k]).max()) for k in sd) / scale
    for k, v in sd.items():
        assert float(np.abs(v.cpu().numpy() - z["sd1_64/" + k]).max()) <= max(1e-5, gap) * scale, (k, gap)
    # captured steps: there from the third epoch on, and the SAME ones after the learning rate changed
    assert seen[-1][0] and all(lr_is_tensor for _, lr_is_tensor in seen[3:])
    changes = np.nonzero(np.diff(h32[:n, 6]))[0]
    if len(changes):
        c = int(changes[0])
        assert c > 5 and seen[c - 1][0] == seen[min(c + 3, n - 1)][0]   # no step was captured anew around the change
    harness.release_steps(r["model"])
