import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True, scope="session")
def _host_threads():
    """The oracle's passes are Python loops of small float64 matmuls: on a many-core GPU host torch's default — one intra-op
    thread per hardware thread (256 on the MI355X boxes) — makes each of them slower, not faster (bench.py's CPU baseline
    calibrates the same thing: 16 threads beat 256 by an order of magnitude on the shape-function loop).  The whole suite's
    host side runs on at most 16 threads."""
    import torch
    before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    yield
    torch.set_num_threads(before)


def golden_names(variant_prefix=None):
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
        names = json.load(f)
    if variant_prefix is None:
        return names
    prefixes = (variant_prefix,) if isinstance(variant_prefix, str) else tuple(variant_prefix)
    return [n for n in names if n.split("_", 2)[2].startswith(prefixes)]


class Golden:
    """One committed fixture: inputs, reference state_dict, fp32/fp64 outputs and gradients."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.meta = json.loads(bytes(z["meta"]).decode())
        if self.meta.get("variant", "").startswith("kink_"):     # inputs exactly on ReLU kinks: same variants, marked
            self.meta["variant"] = self.meta["variant"][5:]
            self.meta["kink"] = True
        self.inputs = {k[3:]: z[k] for k in z.files if k.startswith("in/")}
        self.sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
        self.g32 = {k[4:]: z[k] for k in z.files if k.startswith("g32/")}
        self.g64 = {k[4:]: z[k] for k in z.files if k.startswith("g64/")}
        self.out32 = z["out32"] if "out32" in z.files else None
        self.out64 = z["out64"] if "out64" in z.files else None


@pytest.fixture
def load_golden():
    return Golden
