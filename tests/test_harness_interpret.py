"""f-3 / f-4: the epoch loops against numbers captured from the reference's trainer.py, the interpretability
exports against the oracle.  Plain-torch models: runs on CPU."""
import json
import os

import numpy as np
import pytest
import torch

import gnan_amd  # noqa: F401
from conftest import GOLDEN_DIR, golden_names
from gnan_amd import harness, interpret
from oracle import gnan_oracle as O


class Probe(torch.nn.Module):
    def __init__(self, f, c):
        super().__init__()
        self.lin = torch.nn.Linear(f, c)

    def forward(self, data):
        return self.lin(data.x)


class ToyData:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self


@pytest.mark.parametrize("name", [n for n in golden_names("trainer") if "trainer_gnan" not in n])
def test_epoch_loops_match_reference_trainer(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    batches = []
    for b in range(3):
        batches.append(ToyData(**{k.split("/", 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"b{b}/")}))
    model = Probe(4, meta["C"])
    model.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0/")})
    loss_fn = getattr(torch.nn, meta["loss"])()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    tr = harness.train_epoch(model, batches, loss_fn, opt, "cpu", classify=meta["classify"], compute_auc=False,
                             is_graph_task=meta["graph"])
    assert np.allclose(np.array(tr, dtype=np.float64), z["train_ret"], rtol=1e-5, atol=1e-6)
    for k, v in model.state_dict().items():
        assert np.allclose(v.numpy(), z["sd1/" + k], rtol=1e-5, atol=1e-6), k
    te = harness.test_epoch(model, batches, loss_fn, "cpu", classify=meta["classify"],
                            compute_auc=(meta["loss"] == "BCEWithLogitsLoss" and not meta["graph"]), val_mask=True,
                            is_graph_task=meta["graph"])
    assert np.allclose(np.array(te, dtype=np.float64), z["test_ret"], rtol=1e-5, atol=1e-6)
    assert not model.training                      # trainer.py:97 leaves eval mode on


def test_interpretability_exports_need_the_device():
    """The exports are read off the kernels' tables (tests/test_gpu_harness.py checks them against the oracle): a CPU model
    is refused like every other CPU input, there is no torch re-evaluation to fall back to."""
    from gnan_amd import _lib
    from gnan_amd.models import TensorGNAN
    m = TensorGNAN(4, 2, 3, hidden_channels=8, rho_per_feature=True)
    for call in (lambda: interpret.rho_curve(m, 5), lambda: interpret.shape_functions(m, torch.linspace(-1, 2, 9)),
                 lambda: interpret.contribution_heatmap(m, 5), lambda: interpret.shape_function_tables(m)):
        with pytest.raises(_lib.GnanHipError, match="no CPU fallback"):
            call()


# ---------------------------------------------------------------------------------------------------------- run level
def test_early_stopping_and_loss_rule():
    """main.py:16-41 and main.py:343-352."""
    from gnan_amd.run import CheckpointRules, PatienceCounter, loss_and_out_dim
    pc = PatienceCounter(3)
    for v, stop in ((1.0, False), (0.9, False), (0.95, False), (0.91, False), (0.9, False), (1.2, False), (1.1, False), (1.3, True),
                    (0.1, True)):
        assert pc.observe(v) == stop, v          # 0.9 again is not worse (resets the count); three worse ones in a row stop; it latches
    rules = CheckpointRules(use_auc=False)
    assert rules.due(2.0, 0.7, 0.5, -1) == ["best_val_acc", "best_train_loss"]
    assert rules.due(2.5, 0.6, 0.6, -1) == []                # the bar is now the validation LOSS 0.7 (main.py:200), 0.6 does not beat it
    assert rules.due(1.5, 0.9, 0.8, -1) == ["best_val_acc", "best_train_loss"]
    auc = CheckpointRules(use_auc=True)
    assert auc.due(1.0, 0.5, 0.9, 0.6) == ["best_val_auc", "best_train_loss"] and auc.due(1.0, 0.5, 0.99, 0.6) == []
    assert loss_and_out_dim(2, False) == (torch.nn.BCEWithLogitsLoss, 1)
    assert loss_and_out_dim(40, False) == (torch.nn.CrossEntropyLoss, 40)
    assert loss_and_out_dim(1, True) == (torch.nn.MSELoss, 1)


@pytest.mark.parametrize("name", golden_names("run_exp"))
def test_run_exp_control_flow_follows_the_reference_run(name, tmp_path, monkeypatch):
    """The run loop (scheduler on the training loss, early stopping on the validation loss, the three checkpoint rules and
    their file names, the per-epoch order of passes) fed the per-epoch numbers of the REFERENCE's run (golden 320-321:
    main.py's run_exp driving the reference's classes) must write the same checkpoints at the same epochs, set the same
    learning rates and stop at the same epoch for the same reason."""
    from gnan_amd import run
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    hist = z["hist32"]
    calls = {"train": 0, "eval": []}

    def fake_train(model, dloader, loss_fn, optimizer, classify, device, compute_auc, is_graph_task):
        assert classify is True                     # main.py:159: ~is_regression is truthy whatever the flag
        row = hist[calls["train"]]
        calls["train"] += 1
        return float(row[0]), float(row[1]), -1

    def fake_test(model, dloader, loss_fn, classify, device, compute_auc, is_graph_task, val_mask=False):
        row = hist[calls["train"] - 1]
        calls["eval"].append((calls["train"] - 1, bool(val_mask)))
        return (float(row[2]), float(row[3]), -1) if val_mask else (float(row[4]), float(row[5]), -1)

    monkeypatch.setattr(run.harness, "train_epoch", fake_train)
    monkeypatch.setattr(run.harness, "test_epoch", fake_test)
    model = Probe(4, 2)
    runs = run.run_exp([0], [0], [0], meta["F"], [meta["seed"]], meta["L"], meta["early_stop_flag"], 0.0, "gnan", meta["epochs"],
                       0, meta["wd"], meta["H"], meta["lr"], 1e-5, meta["data_name"], "RUN", int(meta["rho_per_feature"]), 1,
                       meta["graph"], meta["num_classes"], meta["C"], patience=meta["patience"], model=model, device="cpu",
                       checkpoint_dir=str(tmp_path), log=lambda *_: None)
    r = runs[0]
    assert len(r["epochs"]) == meta["epochs_run"] and r["stopped"].startswith(
        {"num_epochs": "num_epochs", "early stop": "early stop at epoch", "loss under": "loss under"}[meta["stopped"]])
    assert [e["lr"] for e in r["epochs"]] == [float(v) for v in hist[:, 6]]
    assert [(e, n) for e, n in r["checkpoints"]] == [tuple(c) for c in meta["checkpoints"]]
    assert sorted(os.listdir(tmp_path)) == sorted({c[1] for c in meta["checkpoints"]})
    sd = torch.load(os.path.join(tmp_path, meta["checkpoints"][-1][1]))
    assert list(sd.keys()) == list(model.state_dict().keys())        # a checkpoint is the state_dict (main.py:172)
    # one validation pass per epoch, a test pass per checkpoint, one more per epoch and one after the loop
    per_epoch = {}
    for e, is_val in calls["eval"]:
        per_epoch.setdefault(e, []).append(is_val)
    assert all(v[0] is True and v.count(True) == 1 for v in per_epoch.values())
    n_ckpt = len(meta["checkpoints"])
    assert sum(not v for _, v in calls["eval"]) == n_ckpt + meta["epochs_run"] + 1
    with pytest.raises(ValueError, match="gnan"):
        run.run_exp([0], [0], [0], 4, [1], 3, 0, 0.0, "gin", 1, 0, 0.0, 8, 0.01, 1e-5, "d", "RUN", 0, 1, False, 2, 1)
