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
