#!/usr/bin/env python3
"""BASELINE config 1 as written: "Cora node-classification via main.py, TensorGNAN on PyTorch CPU (plumbing, no GPU)" — the Cora
SHAPE (2708 nodes, 1433 + 1 features, 7 classes, H = 64, L = 3; the data set itself is not available offline) through the build's
counterpart of main.py (gnan_amd.run.run_exp: GNAN for node tasks, as main.py:79 picks) AND through TensorGNAN, on the CPU:
dense inputs from the build's own preprocessing restatement, the CPU route of gnan_amd/cpu_route.py, stock Adam, the reference's
scheduler / checkpoint / early-stopping rules.  Prints one JSON line.     python tools/c1_cpu.py [epochs]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gnan_amd  # noqa: E402,F401
from gnan_amd import run  # noqa: E402
from gnan_amd.models import TensorGNAN  # noqa: E402
from oracle import gnan_oracle as O  # noqa: E402   (only its preprocessing restatement, to make the dense inputs)

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(0)
n, f_raw, C = 2708, 1433, 7
ei = rng.integers(0, n, (2, 5278))
ei = ei[:, ei[0] != ei[1]]
ei = np.unique(np.concatenate([ei, ei[::-1]], axis=1), axis=1)
nd, norm = O.pre_process_dense(ei, n)
words = (rng.random((n, f_raw)) < 0.0127).astype(np.float32)
words[words.sum(1) == 0, 0] = 1.0
x = torch.from_numpy(np.concatenate([words / words.sum(1, keepdims=True), np.ones((n, 1), np.float32)], axis=1))


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self


perm = rng.permutation(n)
masks = [torch.zeros(n, dtype=torch.bool) for _ in range(3)]
masks[0][perm[:140]] = True
masks[1][perm[140:640]] = True
masks[2][perm[640:1640]] = True
data = Bag(x=x, edge_index=torch.from_numpy(ei), node_distances=nd, normalization_matrix=norm,
           y=torch.from_numpy(rng.integers(0, C, n)), train_mask=masks[0], val_mask=masks[1], test_mask=masks[2])
loader = [data]
out = {"what": "config 1: Cora-shaped node classification on the CPU", "nodes": n, "features": f_raw + 1, "classes": C,
       "threads": torch.get_num_threads(), "epochs": epochs}
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.perf_counter()
    torch.manual_seed(0)
    res = run.run_exp(loader, loader, loader, f_raw + 1, [0], 3, True, 0.0, "gnan", epochs, False, 5e-4, 64, 1e-3, 1e-4, "cora_shaped",
                      "c1", False, True, False, C, C, device=torch.device("cpu"), checkpoint_dir=tmp, log=lambda *_: None)[0]
    out["run_exp_gnan_s_per_epoch"] = (time.perf_counter() - t0) / max(1, len(res["epochs"]))
    out["run_exp_epochs"] = [{k: round(float(v), 6) for k, v in e.items()} for e in res["epochs"]]
    out["checkpoints"] = [name for _, name in res["checkpoints"]][:4]
m = TensorGNAN(f_raw + 1, C, 3, hidden_channels=64)                      # the constructor's defaults: device='cpu'
with torch.no_grad():
    for _, p in m.named_parameters():
        if p.dim() == 2:
            torch.nn.init.xavier_normal_(p, gain=1.0)
t0 = time.perf_counter()
y = m(data)
out["tensor_gnan_forward_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
y.pow(2).mean().backward()
out["tensor_gnan_backward_s"] = time.perf_counter() - t0
sd64 = {k: v.detach().double() for k, v in m.state_dict().items()}
ids = rng.choice(n, 24, replace=False).tolist()
truth = O.gnan_forward(x.double(), nd.double(), norm.double(), sd64, True, node_ids=ids)
out["tensor_gnan_rel_err_vs_float64_oracle_24_rows"] = O.rel_err(y.detach()[ids], truth)
print(json.dumps(out))
