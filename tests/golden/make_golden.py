#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the upstream reference (build container only).

Run from the repo root:  ``python tests/golden/make_golden.py``

The reference lives at /root/reference and never travels: this script imports
its classes in-process (the unused ``torch_geometric`` imports are stubbed in
``sys.modules``), feeds them seeded inputs, and stores *data only* — inputs,
every ``state_dict`` tensor, fp32 outputs, fp64 outputs
(``torch.set_default_dtype(float64)``) and the
gradients of ``out.pow(2).sum()`` — as compressed ``.npz`` files next to this
script.  ``tests/test_oracle_golden.py`` replays them against ``oracle/``.

Weights are re-drawn at O(1) scale before capture: the reference's own
``xavier_normal_(gain=0.01)`` init (GNAN.py:49-53) yields outputs ~1e-14 for
which a relative comparison is meaningless.
"""
import ast
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ import the reference
def _stub_pyg():
    import scipy.sparse

    def to_scipy_sparse_matrix(edge_index, edge_attr=None, num_nodes=None):
        ei = edge_index.cpu().numpy()
        n = int(ei.max()) + 1 if num_nodes is None else num_nodes
        return scipy.sparse.coo_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n))

    pyg = types.ModuleType("torch_geometric")
    nn_ = types.ModuleType("torch_geometric.nn")
    utils = types.ModuleType("torch_geometric.utils")
    for name in ["GraphConv", "GINConv", "GATv2Conv", "GraphSAGE", "TransformerConv", "global_mean_pool"]:
        setattr(nn_, name, object)
    utils.to_scipy_sparse_matrix = to_scipy_sparse_matrix
    pyg.nn, pyg.utils = nn_, utils
    sys.modules.update({"torch_geometric": pyg, "torch_geometric.nn": nn_, "torch_geometric.utils": utils})


_stub_pyg()
sys.path.insert(0, REF)
import GNAN as ref_gnan            # noqa: E402
import models as ref_models        # noqa: E402
import pre_process_datasets as ref_pre  # noqa: E402


def _batched_class():
    """batched_pyg_main.py trains at import; pull out only its model class (lines 98-184)."""
    src = open(os.path.join(REF, "batched_pyg_main.py")).read()
    tree = ast.parse(src)
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TensorGNAN"][0]
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile(ast.Module(body=[node], type_ignores=[]), "batched_pyg_main.py", "exec"), ns)
    return ns["TensorGNAN"]


RefBatched = _batched_class()


class Bag:
    """Duck-typed stand-in for a PyG ``Data`` object (fields read at GNAN.py:56,66,147,161)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


# ------------------------------------------------------------------ inputs
def random_graph(rng, n, directed, n_isolated, avg_deg=2.2):
    """Sparse random graph with >=2 components and isolated nodes so that nd==0 occurs."""
    live = n - n_isolated
    half = max(2, live // 2)
    edges = []
    for lo, hi in [(0, half), (half, live)]:          # two components, never bridged
        size = hi - lo
        if size < 2:
            continue
        m = max(1, int(avg_deg * size / (1 if directed else 2)))
        src = rng.integers(lo, hi, m)
        dst = rng.integers(lo, hi, m)
        keep = src != dst
        edges.append(np.stack([src[keep], dst[keep]]))
    ei = np.concatenate(edges, axis=1)
    ei = np.unique(ei, axis=1)                        # no duplicate edges (they would become weight-2)
    if not directed:
        ei = np.unique(np.concatenate([ei, ei[::-1]], axis=1), axis=1)
    return ei.astype(np.int64)


def run_pre_process(ei, n, f_raw, rng, graph_task, x=None):
    if x is None:
        x = torch.from_numpy(rng.random((n, f_raw), dtype=np.float32))
    data = Bag(x=x, edge_index=torch.from_numpy(ei))
    tmp = tempfile.mkdtemp()
    if graph_task:
        ref_pre.pre_process([data], True, "golden", processed_data_dir=tmp)
    else:
        ref_pre.pre_process(data, False, "golden", processed_data_dir=tmp)
    return data


def redraw(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 2:
                fan_in, fan_out = p.shape[1], p.shape[0]
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (fan_in + fan_out)) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)


def redraw_on_kinks(mode):
    """Weights for the cases whose inputs sit EXACTLY on ReLU kinks (torch: relu'(0) = 0).

    ``zero``: O(1) weights, biases left at the reference's own initial value 0 (GNAN.py:49-53) — every first-layer kink of
    every shape function is at x = 0, and every later pre-activation is 0 there too.
    ``exact``: first-layer weights are multiples of 1/64 and first-layer biases ``-w * a`` with ``a`` in {0, 1/4, 1/2, 1}, so
    the kinks are float32 numbers the inputs take; the other biases are non-zero."""
    def fn(model, seed):
        redraw(model, seed)
        g = torch.Generator().manual_seed(seed + 77)
        with torch.no_grad():
            named = dict(model.named_parameters())
            for name, p in named.items():
                if not name.endswith("bias"):
                    continue
                if mode == "zero":
                    p.zero_()
                elif name.startswith("fs.") and name.split(".")[2] == "0" and p.numel() > 1:
                    w = named[name[:-4] + "weight"]
                    if w.shape[1] != 1:
                        continue
                    q = torch.round(w * 64.0) / 64.0
                    q[q == 0] = 1.0 / 64.0
                    w.copy_(q)
                    a = torch.tensor([0.0, 0.25, 0.5, 1.0])[torch.randint(0, 4, (w.shape[0],), generator=g)]
                    p.copy_(-(w[:, 0] * a))
    return fn


def capture(build, call, data32, seed, redraw_fn=redraw):
    """Run the reference in fp32 and fp64 with identical weights; return arrays to store."""
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(seed)
    m32 = build().eval()
    redraw_fn(m32, seed)
    out32 = call(m32, data32)
    m32.zero_grad()
    out32.pow(2).sum().backward()
    sd = {k: v.detach().clone() for k, v in m32.state_dict().items()}
    g32 = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
           for k, p in m32.named_parameters()}

    torch.set_default_dtype(torch.float64)
    try:
        m64 = build().eval()
        m64.load_state_dict({k: v.double() for k, v in sd.items()})
        data64 = Bag(**{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v)
                        for k, v in data32.__dict__.items()})
        out64 = call(m64, data64)
        m64.zero_grad()
        out64.pow(2).sum().backward()
        g64 = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
               for k, p in m64.named_parameters()}
    finally:
        torch.set_default_dtype(torch.float32)
    arrays = {"out32": out32.detach().numpy(), "out64": out64.detach().numpy()}
    for k, v in sd.items():
        arrays["sd/" + k] = v.numpy()
    for k, v in g32.items():
        arrays["g32/" + k] = v.numpy()
    for k, v in g64.items():
        arrays["g64/" + k] = v.numpy()
    return arrays


def save(name, meta, data, arrays):
    for k in ["x", "edge_index", "node_distances", "normalization_matrix"]:
        if hasattr(data, k):
            arrays["in/" + k] = getattr(data, k).numpy()
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    return os.path.getsize(path)


# ------------------------------------------------------------------ case matrix
def model_cases():
    cases = []
    cid = 0

    def add(variant, **kw):
        nonlocal cid
        base = dict(variant=variant, seed=cid % 3, n=40, f_raw=3, H=8, C=3, L=3, bias=True,
                    normalize_rho=True, directed=False, n_isolated=2)
        base.update(kw)
        base["id"] = cid
        cases.append(base)
        cid += 1

    # A.1 / A.2  stand-alone TensorGNAN (GNAN.py:9-79)
    for L in (1, 2, 3):
        add("standalone_tensor_node", L=L, n=40, C=3)
    add("standalone_tensor_node", normalize_rho=False, C=5, H=32, f_raw=8)
    add("standalone_tensor_node", bias=False, C=1, n=7, n_isolated=1)
    add("standalone_tensor_node", directed=True, n=300, C=3, H=32, f_raw=8, n_isolated=5)
    add("standalone_tensor_graph", C=1, n=40)
    add("standalone_tensor_graph", C=3, n=7, n_isolated=1, L=2)
    add("standalone_tensor_graph", C=1, n=40, normalize_rho=False, directed=True)
    # A.3  models.TensorGNAN (models.py:303-384)
    add("models_tensor_node", rho_per_feature=False, C=3)
    add("models_tensor_node", rho_per_feature=True, C=3)
    add("models_tensor_node", rho_per_feature=True, C=5, n=300, H=32, f_raw=8, n_isolated=5)
    add("models_tensor_node", rho_per_feature=False, C=1, normalize_rho=False, L=2)
    add("models_tensor_node", rho_per_feature=False, C=3, L=1, directed=True)
    add("models_tensor_graph", rho_per_feature=False, C=1, readout_n_layers=0)
    add("models_tensor_graph", rho_per_feature=True, C=3, readout_n_layers=0, n=7, n_isolated=1)
    add("models_tensor_graph", rho_per_feature=False, C=1, readout_n_layers=2)
    add("models_tensor_graph", rho_per_feature=True, C=3, readout_n_layers=2, directed=True)
    add("models_tensor_graph", rho_per_feature=False, C=1, readout_n_layers=1, normalize_rho=False)
    add("models_tensor_graph", rho_per_feature=False, C=1, readout_n_layers=0, bias=False, n=300,
        H=32, f_raw=8, n_isolated=5)
    # A.5  GNAN per-node loop (GNAN.py:82-172, models.py:387-477)
    add("standalone_gnan", rho_per_feature=False, C=3)
    add("standalone_gnan", rho_per_feature=True, C=3, node_ids=[1, 5, 7])
    add("standalone_gnan", rho_per_feature=False, C=1, normalize_rho=False, L=2)
    add("models_gnan", rho_per_feature=False, C=3, node_ids=[1, 5, 7])
    add("models_gnan", rho_per_feature=True, C=5, H=32, f_raw=8)
    add("models_gnan", rho_per_feature=False, C=3, L=1, n=7, n_isolated=1)
    add("models_gnan", rho_per_feature=False, C=1, n=300, H=32, f_raw=8, n_isolated=5, directed=True)
    # NAM (models.py:259-300)
    add("models_nam", C=3, L=3)
    add("models_nam", C=1, L=2, bias=False)
    return cases


def build_and_call(c, F):
    v = c["variant"]
    kw = dict(in_channels=F, out_channels=c["C"], hidden_channels=c["H"], bias=c["bias"], dropout=0.0,
              device="cpu")
    if v.startswith("standalone_tensor"):
        graph = v.endswith("graph")
        return (lambda: ref_gnan.TensorGNAN(n_layers=c["L"], normalize_rho=c["normalize_rho"],
                                            is_graph_task=graph, **kw),
                lambda m, d: m.forward(d))
    if v.startswith("models_tensor"):
        graph = v.endswith("graph")
        return (lambda: ref_models.TensorGNAN(n_layers=c["L"], normalize_rho=c["normalize_rho"],
                                              is_graph_task=graph, rho_per_feature=c["rho_per_feature"],
                                              readout_n_layers=c.get("readout_n_layers", 0), **kw),
                lambda m, d: m.forward(d))
    if v == "standalone_gnan":
        return (lambda: ref_gnan.GNAN(n_layers=c["L"], normalize_rho=c["normalize_rho"],
                                      rho_per_feature=c["rho_per_feature"], **kw),
                lambda m, d: m.forward(d, c.get("node_ids")))
    if v == "models_gnan":
        return (lambda: ref_models.GNAN(num_layers=c["L"], normalize_rho=c["normalize_rho"],
                                        rho_per_feature=c["rho_per_feature"], **kw),
                lambda m, d: m.forward(d, c.get("node_ids")))
    if v == "models_nam":
        return (lambda: ref_models.NAM(num_layers=c["L"], **kw), lambda m, d: m.forward(d.x))
    raise ValueError(v)


# f-3 with the path itself: the reference's epoch loops driving the reference's GNAN classes (trainer.py:23-154 over
# models.py:304-477 / GNAN.py:9-79) — one SGD epoch and one evaluation pass; inputs, initial and updated state_dict
class GraphData(Bag):
    def to(self, device):
        return self


def trainer_case(manifest, tid, variant, C, loss_name, n_batches, n, f_raw=4, H=8, L=3, opt_name="SGD", epochs=1,
                 with_f64=False):
    """``with_f64`` (make_golden_run.py, the Adam cases): the same run once more under ``torch.set_default_dtype(float64)``
    from the SAME data and initial weights — ``train_hist64`` / ``test_hist64`` / ``sd1_64``: the gap to the float32 run is
    the reference's own float32 trajectory error, the bound the GPU harness is held to."""
    import trainer as ref_trainer
    graph_task = variant.endswith("graph")
    rng = np.random.default_rng(9100 + tid)
    batches = []
    for b in range(n_batches):
        nb = n if not graph_task else int(rng.integers(5, 12))
        ei = random_graph(rng, nb, False, 1)
        if not graph_task:
            perm = np.arange(nb)
            top = int(ei.max())
            perm[[top, nb - 1]] = perm[[nb - 1, top]]
            ei = perm[ei]
        d = run_pre_process(ei, nb, f_raw, rng, graph_task)
        ny = 1 if graph_task else nb
        if loss_name == "CrossEntropyLoss":
            y = torch.from_numpy(rng.integers(0, C, ny))
        elif loss_name == "MSELoss":
            y = torch.from_numpy(rng.standard_normal(ny).astype(np.float32))
        else:
            y = torch.from_numpy(rng.choice([-1.0, 1.0], ny).astype(np.float32))
        fields = dict(x=d.x, edge_index=d.edge_index, node_distances=d.node_distances,
                      normalization_matrix=d.normalization_matrix, y=y)
        if not graph_task:
            for m in ("train_mask", "val_mask", "test_mask"):
                mask = torch.from_numpy(rng.random(nb) < 0.6)
                mask[0] = True
                fields[m] = mask
        batches.append(GraphData(**fields))
    F = batches[0].x.shape[1]
    c = dict(variant=variant, C=C, H=H, L=L, bias=True, normalize_rho=True, rho_per_feature=(C > 1),
             readout_n_layers=0)
    build, _ = build_and_call(c, F)
    torch.manual_seed(60 + tid)
    model = build()
    redraw(model, 60 + tid)
    model.train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    loss_fn = getattr(torch.nn, loss_name)()
    lr = 0.05 if opt_name == "SGD" else 0.01
    opt = torch.optim.SGD(model.parameters(), lr=lr) if opt_name == "SGD" else torch.optim.Adam(model.parameters(), lr=lr)
    classify = loss_name != "MSELoss"
    hist_tr, hist_te = [], []
    for _ in range(epochs):               # main.py:176-215: a training pass, then an evaluation pass, every epoch
        tr = ref_trainer.train_epoch(model, batches, loss_fn, opt, "cpu", classify=classify, compute_auc=False,
                                     is_graph_task=graph_task)
        sd1 = {k: v.clone() for k, v in model.state_dict().items()}
        te = ref_trainer.test_epoch(model, batches, loss_fn, "cpu", classify=classify, compute_auc=False,
                                    val_mask=True, is_graph_task=graph_task)
        hist_tr.append(tr)
        hist_te.append(te)
    arrays = {"train_ret": np.array(tr, dtype=np.float64), "test_ret": np.array(te, dtype=np.float64)}
    if epochs > 1:
        arrays["train_hist"] = np.array(hist_tr, dtype=np.float64)
        arrays["test_hist"] = np.array(hist_te, dtype=np.float64)
    if with_f64:
        real_bce, real_mse = torch.nn.BCEWithLogitsLoss.forward, torch.nn.MSELoss.forward
        torch.nn.BCEWithLogitsLoss.forward = lambda self, inp, target: real_bce(self, inp, target.to(inp.dtype))   # trainer.py:62 hands float32 labels over
        torch.nn.MSELoss.forward = lambda self, inp, target: real_mse(self, inp, target.to(inp.dtype))
        try:
            m64 = build()
            m64.load_state_dict(sd0)
            m64.double().train()
            torch.set_default_dtype(torch.float64)            # GNAN.py:57 / models.py:360 allocate fx in the DEFAULT dtype
            b64 = [GraphData(**{k: (v.double() if v.is_floating_point() else v) for k, v in d.__dict__.items()}) for d in batches]
            opt64 = torch.optim.SGD(m64.parameters(), lr=lr) if opt_name == "SGD" else torch.optim.Adam(m64.parameters(), lr=lr)
            h_tr, h_te = [], []
            for _ in range(epochs):
                h_tr.append(ref_trainer.train_epoch(m64, b64, loss_fn, opt64, "cpu", classify=classify, compute_auc=False,
                                                    is_graph_task=graph_task))
                h_te.append(ref_trainer.test_epoch(m64, b64, loss_fn, "cpu", classify=classify, compute_auc=False,
                                                   val_mask=True, is_graph_task=graph_task))
        finally:
            torch.set_default_dtype(torch.float32)
            torch.nn.BCEWithLogitsLoss.forward = real_bce
            torch.nn.MSELoss.forward = real_mse
        arrays["train_hist64"] = np.array(h_tr, dtype=np.float64)
        arrays["test_hist64"] = np.array(h_te, dtype=np.float64)
        for k, v in m64.state_dict().items():
            arrays["sd1_64/" + k] = v.numpy()
    for k, v in sd0.items():
        arrays["sd0/" + k] = v.numpy()
    for k, v in sd1.items():
        arrays["sd1/" + k] = v.numpy()
    for b, d in enumerate(batches):
        for k, v in d.__dict__.items():
            arrays[f"b{b}/{k}"] = v.numpy()
    meta = dict(variant="trainer_gnan", model=variant, id=310 + tid, C=C, H=H, L=L, F=F, graph=graph_task, loss=loss_name,
                classify=classify, n_batches=n_batches, lr=lr, rho_per_feature=(C > 1), optimizer=opt_name, epochs=epochs)
    name = f"case_{310 + tid:03d}_trainer_gnan"
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    if name not in manifest:
        manifest.append(name)
    print(name, tr, te)


def main():
    total = 0
    manifest = []
    for c in model_cases():
        rng = np.random.default_rng(1000 + c["id"])
        graph_task = c["variant"].endswith("graph")
        ei = random_graph(rng, c["n"], c["directed"], c["n_isolated"])
        if not graph_task:
            # node-task preprocessing infers N from the largest edge endpoint (pre_process_datasets.py:128):
            # keep the isolated nodes in the middle of the id range, not at the end.
            n = c["n"]
            perm = np.arange(n)
            top = int(ei.max())
            perm[[top, n - 1]] = perm[[n - 1, top]]
            ei = perm[ei]
        data = run_pre_process(ei, c["n"], c["f_raw"], rng, graph_task)
        F = data.x.shape[1]
        build, call = build_and_call(c, F)
        arrays = capture(build, call, data, c["seed"])
        name = f"case_{c['id']:03d}_{c['variant']}"
        size = save(name, c, data, arrays)
        total += size
        manifest.append(name)
        print(f"{name}: out {arrays['out32'].shape} {size / 1024:.1f} KiB")

    # f-2: batched block-diagonal variant (batched_pyg_main.py:98-184)
    for bid, (C, graph) in enumerate([(2, True), (3, False)]):
        rng = np.random.default_rng(5000 + bid)
        sizes = [5, 9, 4, 12]
        xs, blocks = [], []
        for s in sizes:
            ei = random_graph(rng, s, False, 1 if s > 4 else 0)
            d = run_pre_process(ei, s, 3, rng, True)
            hop = torch.round(1.0 / d.node_distances.clamp_min(1e-9)) - 1.0
            hop[d.node_distances == 0] = -1.0        # unreachable marked like cross-graph pairs
            xs.append(d.x)
            blocks.append(hop)
        n = sum(sizes)
        dist = torch.full((n, n), -1.0)
        o = 0
        for s, b in zip(sizes, blocks):
            dist[o:o + s, o:o + s] = b
            o += s
        batch = torch.cat([torch.full((s,), i, dtype=torch.long) for i, s in enumerate(sizes)])
        data = Bag(x=torch.cat(xs), dist=dist, batch=batch)
        F = data.x.shape[1]
        c = dict(variant="batched_tensor", id=100 + bid, seed=bid, C=C, H=8, F=F, graph=graph)
        arrays = capture(lambda: RefBatched(F, C, 2, hidden_channels=8, is_graph_task=graph),
                         lambda m, d: m.forward(d.x, d.dist, d.batch), data, bid)
        arrays["in/x"], arrays["in/dist"], arrays["in/batch"] = data.x.numpy(), dist.numpy(), batch.numpy()
        name = f"case_{100 + bid:03d}_batched_tensor"
        total += save(name, c, Bag(), arrays)
        manifest.append(name)
        print(name)

    # f-1: preprocessing in -> out pairs (pre_process_datasets.py:104-142)
    for pid, (n, directed, iso) in enumerate([(7, False, 1), (12, True, 2), (60, False, 4), (25, True, 0)]):
        rng = np.random.default_rng(7000 + pid)
        ei = random_graph(rng, n, directed, iso)
        d = run_pre_process(ei, n, 2, rng, True)
        c = dict(variant="pre_process", id=200 + pid, n=n, directed=directed)
        name = f"case_{200 + pid:03d}_pre_process"
        total += save(name, c, d, {})
        manifest.append(name)
        print(name)

    # f-3: the reference's epoch loops on a plain torch model (trainer.py:23-154)
    import trainer as ref_trainer                     # noqa: E402  (sklearn only; importable here)

    class Probe(torch.nn.Module):                     # forward(data) -> logits, like the GNAN classes
        def __init__(self, f, c):
            super().__init__()
            self.lin = torch.nn.Linear(f, c)

        def forward(self, data):
            return self.lin(data.x)

    class ToyData(Bag):
        def to(self, device):
            return self

    for tid, (c, graph_task, loss_name) in enumerate([(1, False, "BCEWithLogitsLoss"), (3, False, "CrossEntropyLoss"),
                                                      (1, True, "MSELoss"), (1, True, "BCEWithLogitsLoss")]):
        torch.manual_seed(40 + tid)
        rng = np.random.default_rng(9000 + tid)
        batches = []
        for b in range(3):
            n = 12 if not graph_task else 1
            x = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
            if loss_name == "CrossEntropyLoss":
                y = torch.from_numpy(rng.integers(0, c, n))
            elif loss_name == "MSELoss":
                y = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
            else:
                y = torch.from_numpy(rng.choice([-1.0, 1.0], n).astype(np.float32))   # {-1,+1}: remapped to {0,1}
            masks = {m: torch.from_numpy(rng.random(n) < 0.6) for m in ("train_mask", "val_mask", "test_mask")}
            for m in masks.values():
                m[0] = True
            batches.append(ToyData(x=x, y=y, **masks))
        model = Probe(4, c)
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        loss_fn = getattr(torch.nn, loss_name)()
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        classify = loss_name != "MSELoss"
        tr = ref_trainer.train_epoch(model, batches, loss_fn, opt, "cpu", classify=classify, compute_auc=False,
                                     is_graph_task=graph_task)
        sd1 = {k: v.clone() for k, v in model.state_dict().items()}
        te = ref_trainer.test_epoch(model, batches, loss_fn, "cpu", classify=classify,
                                    compute_auc=(loss_name == "BCEWithLogitsLoss" and not graph_task),
                                    val_mask=True, is_graph_task=graph_task)
        arrays = {"train_ret": np.array(tr, dtype=np.float64), "test_ret": np.array(te, dtype=np.float64)}
        for k, v in sd0.items():
            arrays["sd0/" + k] = v.numpy()
        for k, v in sd1.items():
            arrays["sd1/" + k] = v.numpy()
        for b, d in enumerate(batches):
            for k, v in d.__dict__.items():
                arrays[f"b{b}/{k}"] = v.numpy()
        meta = dict(variant="trainer", id=300 + tid, C=c, graph=graph_task, loss=loss_name, classify=classify)
        name = f"case_{300 + tid:03d}_trainer"
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        manifest.append(name)
        print(name, tr, te)

    trainer_case(manifest, 0, "models_tensor_node", 3, "CrossEntropyLoss", 1, 40)
    trainer_case(manifest, 1, "models_gnan", 1, "BCEWithLogitsLoss", 1, 30)
    trainer_case(manifest, 2, "models_tensor_graph", 1, "BCEWithLogitsLoss", 4, 0)
    trainer_case(manifest, 3, "standalone_tensor_node", 3, "CrossEntropyLoss", 1, 40)
    trainer_case(manifest, 4, "models_tensor_graph", 1, "MSELoss", 3, 0)
    # Adam over several epochs (main.py:141): long enough for the harness to capture the step and replay it
    trainer_case(manifest, 5, "models_tensor_node", 3, "CrossEntropyLoss", 1, 60, f_raw=6, H=16, opt_name="Adam", epochs=6, with_f64=True)
    trainer_case(manifest, 6, "models_tensor_graph", 1, "BCEWithLogitsLoss", 3, 0, opt_name="Adam", epochs=6, with_f64=True)
    trainer_case(manifest, 7, "standalone_tensor_node", 2, "CrossEntropyLoss", 1, 50, opt_name="Adam", epochs=5, with_f64=True)

    # inputs EXACTLY on ReLU kinks: zero biases (the reference's own initial state) with one-hot / bag-of-words style
    # features, and kinks placed on float32 numbers the inputs take.  torch differentiates relu at 0 as 0.
    kink_cases = [
        dict(variant="models_tensor_node", mode="zero", n=40, f_raw=5, H=8, C=3, L=3, rho_per_feature=True),
        dict(variant="standalone_tensor_node", mode="zero", n=300, f_raw=8, H=32, C=1, L=3),
        dict(variant="models_gnan", mode="zero", n=40, f_raw=5, H=8, C=3, L=2, rho_per_feature=False),
        dict(variant="models_tensor_node", mode="exact", n=300, f_raw=8, H=16, C=2, L=3, rho_per_feature=False),
        dict(variant="models_tensor_graph", mode="zero", n=40, f_raw=5, H=8, C=1, L=3, rho_per_feature=False,
             readout_n_layers=0),
        dict(variant="models_tensor_node", mode="exact", n=40, f_raw=5, H=8, C=1, L=2, rho_per_feature=False),
    ]
    for kid, kc in enumerate(kink_cases):
        c = dict(seed=kid % 3, bias=True, normalize_rho=True, directed=False, n_isolated=2, id=400 + kid)
        c.update(kc)
        rng = np.random.default_rng(11000 + kid)
        graph_task = c["variant"].endswith("graph")
        ei = random_graph(rng, c["n"], False, c["n_isolated"])
        if not graph_task:
            perm = np.arange(c["n"])
            top = int(ei.max())
            perm[[top, c["n"] - 1]] = perm[[c["n"] - 1, top]]
            ei = perm[ei]
        levels = np.array([0.0, 0.0, 0.0, 1.0, 1.0, 0.25, 0.5, 0.75], dtype=np.float32)
        x = torch.from_numpy(levels[rng.integers(0, len(levels), (c["n"], c["f_raw"]))])
        data = run_pre_process(ei, c["n"], c["f_raw"], rng, graph_task, x=x)
        build, call = build_and_call(c, data.x.shape[1])
        arrays = capture(build, call, data, c["seed"], redraw_fn=redraw_on_kinks(c["mode"]))
        name = f"case_{400 + kid:03d}_kink_{c['variant']}"
        c["variant"] = "kink_" + c["variant"]
        total += save(name, c, data, arrays)
        manifest.append(name)
        print(f"{name}: out {arrays['out32'].shape}")

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(f"{len(manifest)} cases, {total / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
