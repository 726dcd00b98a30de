"""GPU parity tests proper: the HIP path (through the C ABI) vs the golden vectors and the oracle.

Tolerance: truth is the reference run in float64; the build passes iff
``max|y - y64| / max|y64| <= max(1e-5, the same quantity for the fp32 reference)``.
Integer / index outputs (hop codes, shell counts) must be bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import Golden, golden_names
from helpers import TWO_FLOORS, grad_rule, inputs_from, module_grads, oracle_grads, params_from, tolerance_ok
from oracle import gnan_oracle as O
from gnan_amd import aggregate  # noqa: E402  (the aggregation's switches are patched below)

pytestmark = pytest.mark.gpu

MODEL_CASES = [n for n in golden_names() if not any(t in n for t in ("pre_process", "batched", "trainer", "run_exp"))]


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    import gpu_util
    return gpu_util


@pytest.fixture(params=["auto", "pwl"])
def strategy(request, monkeypatch):
    """Shape-function strategy: what AUTO picks at these sizes (matrix-core / lane kernel) and the forced
    piecewise-linear table path (what AUTO picks for large graphs)."""
    from gnan_amd import _lib, functional
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_AUTO if request.param == "auto" else _lib.FMLP_PWL)
    return request.param


@pytest.mark.parametrize("name", MODEL_CASES)
def test_forward_matches_golden(gpu, name, strategy):
    g = Golden(name)
    mod = gpu.build_module(g)
    with torch.no_grad():
        y = gpu.call(mod, g, gpu.device_inputs(g)).cpu()
    assert tuple(y.shape) == g.out32.shape
    ok, e_build, e_ref = tolerance_ok(y, g.out32, g.out64, floor=1e-5)
    assert ok, f"build err {e_build:.3e} vs fp32-reference err {e_ref:.3e}"


@pytest.mark.parametrize("name", MODEL_CASES)
def test_backward_matches_golden(gpu, name, strategy):
    g = Golden(name)
    mod = gpu.build_module(g)
    y = gpu.call(mod, g, gpu.device_inputs(g))
    y.pow(2).sum().backward()
    named = dict(mod.named_parameters())
    ok, e_build, e_ref, where = grad_rule({k: named[k].grad for k in g.g64}, g.g64, g.g32)      # SURVEY 8c, on the gradient
    assert ok, f"{where}: build {e_build:.3e} vs fp32-reference {e_ref:.3e}"


@pytest.mark.parametrize("name,use_cnt", [("case_004_standalone_tensor_node", "pre"), ("case_006_standalone_tensor_graph", "pre"),
                                          ("case_008_standalone_tensor_graph", False), ("case_014_models_tensor_graph", True)])
def test_small_golden_graphs_take_the_one_launch_path(gpu, name, use_cnt, monkeypatch):
    """The golden cases of small graphs with one output channel — the stand-alone file's pre-rho normalisation included — run
    through gnan_small_graph_fwd / _bwd (test_forward_matches_golden / test_backward_matches_golden above compare the values)."""
    from gnan_amd import small_graph
    g = Golden(name)
    mod = gpu.build_module(g)
    seen = []
    real = small_graph.small_graph_forward
    monkeypatch.setattr(small_graph, "small_graph_forward", lambda x, hg, f, r, uc, gs: seen.append(uc) or real(x, hg, f, r, uc, gs))
    y = gpu.call(mod, g, gpu.device_inputs(g))
    assert seen == [use_cnt]
    ok, e_build, e_ref = tolerance_ok(y.detach().cpu(), g.out32, g.out64, floor=1e-5)
    assert ok, f"build err {e_build:.3e} vs fp32-reference err {e_ref:.3e}"
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        y.pow(2).sum().backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages() if "small_graph_bwd_kernel" in e.key]
    assert names, [e.key for e in prof.key_averages()]


@pytest.mark.parametrize("name", ["case_016_models_tensor_graph", "case_017_models_tensor_graph", "case_018_models_tensor_graph"])
def test_nam_readout_goldens_take_the_one_launch_path(gpu, name, monkeypatch):
    """Golden cases 016-018 (graph task with a NAM read-out of one or two layers, one and three classes, with and without
    normalisation) run through gnan_small_graph_nam_fwd / _bwd: outputs and every parameter gradient vs the reference's."""
    from gnan_amd import small_graph
    g = Golden(name)
    mod = gpu.build_module(g)
    seen = []
    real = small_graph.small_graph_nam_forward
    monkeypatch.setattr(small_graph, "small_graph_nam_forward", lambda *a: seen.append(1) or real(*a))
    y = gpu.call(mod, g, gpu.device_inputs(g))
    assert seen == [1]
    ok, e_build, e_ref = tolerance_ok(y.detach().cpu(), g.out32, g.out64, floor=1e-5)
    assert ok, f"build err {e_build:.3e} vs fp32-reference err {e_ref:.3e}"
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        y.pow(2).sum().backward()
        torch.cuda.synchronize()
    assert any("small_graph_nam_bwd_kernel" in e.key for e in prof.key_averages())
    named = dict(mod.named_parameters())
    ok, e_build, e_ref, where = grad_rule({k: named[k].grad for k in g.g64}, g.g64, g.g32)      # SURVEY 8c, on the gradient
    assert ok, f"{where}: build {e_build:.3e} vs fp32-reference {e_ref:.3e}"


@pytest.mark.parametrize("name", golden_names(("standalone_tensor_node", "models_tensor_node", "models_gnan")))
@pytest.mark.parametrize("K", [1, 2])
def test_truncated_csr_matches_dense_definition(gpu, name, K):
    """A K-hop CSR is *defined* as the reference on the dense input with hops > K zeroed."""
    from gnan_amd import HopGraph
    g = Golden(name)
    m = g.meta
    if m.get("node_ids"):
        pytest.skip("row subsets are covered by the dense golden test")
    i64, p64 = inputs_from(g, torch.float64), params_from(g, torch.float64)
    i32, p32 = inputs_from(g, torch.float32), params_from(g, torch.float32)
    nd_k, norm_k = O.truncate_dense(i32["node_distances"], K)
    pre_rho = m["variant"].startswith("standalone_tensor")
    fwd = O.tensor_gnan_forward_standalone if pre_rho else O.tensor_gnan_forward_models
    truth = fwd(i64["x"], nd_k.double(), norm_k.double(), p64, m["normalize_rho"], False)
    ref32 = fwd(i32["x"], nd_k, norm_k, p32, m["normalize_rho"], False)
    hops = O.hop_codes_from_dense(nd_k)
    rowptr, col, code = O.csr_from_hops(hops, K)
    graph = HopGraph.from_csr(torch.from_numpy(rowptr).to(gpu.DEV), torch.from_numpy(col).to(gpu.DEV),
                              torch.from_numpy(code).to(gpu.DEV), n_cols=hops.shape[1], n_codes=K + 2)
    assert np.array_equal(graph.cnt.cpu().numpy(), O.shell_counts(hops, K + 2))      # bit-exact index work
    mod = gpu.build_module(g)
    data = gpu.Bag(x=i32["x"].to(gpu.DEV), edge_index=None, gnan_graph=graph)
    with torch.no_grad():
        y = mod.forward(data).cpu()
    ok, e_build, e_ref = tolerance_ok(y, ref32, truth, floor=1e-5)
    assert ok, f"K={K}: build err {e_build:.3e} vs fp32 err {e_ref:.3e}"


@pytest.mark.parametrize("name", golden_names("pre_process") + [MODEL_CASES[5], MODEL_CASES[11]])
def test_dense_to_code_bit_exact(gpu, name):
    from gnan_amd import HopGraph
    g = Golden(name)
    nd = torch.from_numpy(np.array(g.inputs["node_distances"]))
    norm = torch.from_numpy(np.array(g.inputs["normalization_matrix"]))
    graph = HopGraph.from_dense(nd.to(gpu.DEV), norm.to(gpu.DEV))
    hops = O.hop_codes_from_dense(nd)
    D = int(hops.max()) + 2
    assert graph.n_codes == D
    want = np.where(hops < 0, 255, hops).astype(np.uint8)
    assert np.array_equal(graph.code.cpu().numpy(), want)
    assert np.array_equal(graph.cnt.cpu().numpy(), O.shell_counts(hops, D))


def test_dense_to_code_rejects_foreign_inputs(gpu):
    from gnan_amd import HopGraph
    from gnan_amd._lib import GnanHipError
    nd = torch.tensor([[1.0, 0.5, 0.3], [0.5, 1.0, 0.5], [0.0, 0.5, 1.0]], device=gpu.DEV)   # 0.3 is no 1/(1+hop)
    with pytest.raises(GnanHipError, match="node_distances"):
        HopGraph.from_dense(nd)
    nd = torch.tensor([[1.0, 0.5], [0.5, 1.0]], device=gpu.DEV)
    with pytest.raises(GnanHipError, match="normalization_matrix"):
        HopGraph.from_dense(nd, torch.full((2, 2), 2.0, device=gpu.DEV))


@pytest.mark.parametrize("name", golden_names("models_tensor_node"))
def test_reference_order_and_feature_contributions(gpu, name):
    """aggregation_order='reference' (models.py:373-376: aggregate F*C columns, then sum over features) and the
    per-feature contribution tensor mf give the same outputs as the sum-first default."""
    g = Golden(name)
    mod = gpu.build_module(g)
    data = gpu.device_inputs(g)
    mod.aggregation_order = "reference"
    with torch.no_grad():
        y = mod.forward(data).cpu()
        mf = mod.feature_contributions(data).cpu()               # [N, F, C]
    ok, e_build, e_ref = tolerance_ok(y, g.out32, g.out64, floor=1e-5)
    assert ok, f"reference order: build err {e_build:.3e} vs fp32-reference err {e_ref:.3e}"
    ok, e_build, e_ref = tolerance_ok(mf.sum(1), g.out32, g.out64, floor=1e-5)
    assert ok, f"feature_contributions: build err {e_build:.3e}"
    # truth for the contribution tensor itself: the reference's mf = m @ fx, from the oracle pieces in float64
    i64, p64 = inputs_from(g, torch.float64), params_from(g, torch.float64)
    fx = O.feature_mlps(i64["x"], p64)
    m = O._rho_dense(i64["node_distances"], p64)
    if g.meta["normalize_rho"]:
        m = m / i64["normalization_matrix"].unsqueeze(-1)
    mf64 = torch.matmul(m.permute(2, 0, 1), fx.permute(2, 0, 1)).permute(1, 2, 0)   # [N, F, C]
    assert O.rel_err(mf, mf64) <= 1e-5


@pytest.mark.parametrize("name", golden_names("batched"))
def test_batched_variant_matches_golden(gpu, name):
    """f-2: the block-diagonal batched TensorGNAN (batched_pyg_main.py:98-184), forward and gradients."""
    from gnan_amd.batched import TensorGNAN
    g = Golden(name)
    m = g.meta
    mod = TensorGNAN(m["F"], m["C"], 2, hidden_channels=m["H"], is_graph_task=m["graph"], device=gpu.DEV)
    mod.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in g.sd.items()}, strict=True)
    mod = mod.to(gpu.DEV).eval()
    x = torch.from_numpy(np.array(g.inputs["x"])).to(gpu.DEV)
    dist = torch.from_numpy(np.array(g.inputs["dist"])).to(gpu.DEV)
    batch = torch.from_numpy(np.array(g.inputs["batch"])).to(gpu.DEV)
    y = mod.forward(x, dist, batch)
    assert tuple(y.shape) == g.out32.shape
    ok, e_build, e_ref = tolerance_ok(y.detach().cpu(), g.out32, g.out64, floor=1e-5)
    assert ok, f"build err {e_build:.3e} vs fp32-reference err {e_ref:.3e}"
    y.pow(2).sum().backward()
    named = dict(mod.named_parameters())
    ok, e_build, e_ref, where = grad_rule({k: named[k].grad for k in g.g64}, g.g64, g.g32)      # SURVEY 8c, on the gradient
    assert ok, f"{where}: build {e_build:.3e} vs fp32-reference {e_ref:.3e}"


@pytest.mark.parametrize("name", golden_names("pre_process") + [MODEL_CASES[5], MODEL_CASES[11], MODEL_CASES[0]])
@pytest.mark.parametrize("max_hops", [None, 2, 1])
def test_gpu_preprocessing_bit_exact(gpu, name, max_hops):
    """f-1: hop codes / shell counts from edge_index on the GPU == what the reference's pre_process encodes."""
    from gnan_amd import HopGraph
    g = Golden(name)
    n = g.inputs["node_distances"].shape[0]
    ei = torch.from_numpy(np.array(g.inputs["edge_index"])).to(gpu.DEV)
    hops = O.hop_codes_from_dense(torch.from_numpy(np.array(g.inputs["node_distances"])))
    graph = HopGraph.from_edge_index(ei, n, max_hops)
    if max_hops == 1:
        rowptr, col, code = O.csr_from_hops(hops, 1)
        assert np.array_equal(graph.rowptr.cpu().numpy(), rowptr)
        assert np.array_equal(graph.col.cpu().numpy(), col)
        assert np.array_equal(graph.code.cpu().numpy(), code)
        assert np.array_equal(graph.cnt.cpu().numpy(), O.shell_counts(np.where(hops > 1, -1, hops), 3))
        return
    if max_hops is not None:
        hops = np.where(hops > max_hops, -1, hops)
    D = int(hops.max()) + 2
    assert graph.n_codes == D
    assert np.array_equal(graph.code.cpu().numpy(), np.where(hops < 0, 255, hops).astype(np.uint8))
    assert np.array_equal(graph.cnt.cpu().numpy(), O.shell_counts(hops, D))
    if max_hops is None:                                      # and it is interchangeable with the dense-input route
        dense = HopGraph.from_dense(torch.from_numpy(np.array(g.inputs["node_distances"])).to(gpu.DEV),
                                    torch.from_numpy(np.array(g.inputs["normalization_matrix"])).to(gpu.DEV))
        assert torch.equal(dense.code, graph.code) and torch.equal(dense.cnt, graph.cnt)


def _khop_expected(hops, K, lo, hi):
    """(rowptr, col, code) with rows [lo, hi) ordered by (hop, node id) — the order gnan_amd documents."""
    cols, codes, ptr = [], [], [0]
    for i in range(lo, hi):
        j = np.nonzero((hops[i] >= 0) & (hops[i] <= K))[0]
        order = np.lexsort((j, hops[i, j]))
        cols.append(j[order])
        codes.append(hops[i, j[order]])
        ptr.append(ptr[-1] + len(j))
    return np.array(ptr), np.concatenate(cols).astype(np.int32), np.concatenate(codes).astype(np.uint8)


@pytest.mark.parametrize("name", golden_names("pre_process") + [MODEL_CASES[5], MODEL_CASES[11]])
@pytest.mark.parametrize("K", [2, 3, 6])
def test_gpu_khop_csr_preprocessing_bit_exact(gpu, name, K, monkeypatch):
    """f-1 for graphs too large for N^2 bytes: K-hop truncated hop-coded CSR straight from edge_index
    (gnan_bfs_khop) == the reference's pre_process matrices with everything beyond K hops zeroed."""
    from gnan_amd import HopGraph, graph as graph_mod
    g = Golden(name)
    n = g.inputs["node_distances"].shape[0]
    ei = torch.from_numpy(np.array(g.inputs["edge_index"])).to(gpu.DEV)
    hops = O.hop_codes_from_dense(torch.from_numpy(np.array(g.inputs["node_distances"])))
    monkeypatch.setattr(graph_mod, "KHOP_QUEUE_START", 2)          # also walks the queue-growth path
    got = HopGraph.from_edge_index(ei, n, K, layout="csr")
    rowptr, col, code = _khop_expected(hops, K, 0, n)
    assert got.n_codes == K + 2 and got.n_cols == n
    assert np.array_equal(got.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(got.col.cpu().numpy(), col)
    assert np.array_equal(got.code.cpu().numpy(), code)
    assert np.array_equal(got.cnt.cpu().numpy(), O.shell_counts(np.where(hops > K, -1, hops), K + 2))
    lo, hi = n // 3, n - 1                                         # a row block keeps global column ids
    blk = HopGraph.from_edge_index(ei, n, K, layout="csr", rows=(lo, hi))
    rp, cb, cd = _khop_expected(hops, K, lo, hi)
    assert np.array_equal(blk.rowptr.cpu().numpy(), rp) and np.array_equal(blk.col.cpu().numpy(), cb)
    assert np.array_equal(blk.code.cpu().numpy(), cd)
    assert torch.equal(blk.cnt, got.cnt[lo:hi])


@pytest.mark.parametrize("name", [MODEL_CASES[9], MODEL_CASES[11]])
def test_khop_csr_forward_matches_truncated_dense_reference(gpu, name):
    """Module forward on the K = 2 CSR built on the device == oracle on the dense inputs truncated at 2 hops."""
    from gnan_amd import HopGraph
    g = Golden(name)
    if g.meta.get("node_ids") or not g.meta["variant"].startswith("models_tensor_node"):
        pytest.skip("node-level models.TensorGNAN fixtures only")
    i64 = inputs_from(g, torch.float64)
    p64 = params_from(g, torch.float64)
    nd_k, norm_k = O.truncate_dense(i64["node_distances"], 2)
    truth = O.tensor_gnan_forward_models(i64["x"], nd_k, norm_k, p64, g.meta["normalize_rho"], False)
    mod = gpu.build_module(g)
    data = gpu.device_inputs(g)
    n = data.x.shape[0]
    data.gnan_graph = HopGraph.from_edge_index(torch.from_numpy(np.array(g.inputs["edge_index"])).to(gpu.DEV), n, 2,
                                               layout="csr")
    with torch.no_grad():
        y = mod.forward(data)
    assert O.rel_err(y.cpu().double(), truth) <= 1e-5


def test_training_mode_dropout_is_supported(gpu):
    """run.sh trains with dropout 0.6 and the module starts in train mode: the first epoch must work.
    Dropout is stochastic, so the check is determinism under a seed, finiteness, and that eval() is unaffected."""
    import warnings
    g = Golden(MODEL_CASES[9])                                  # models.TensorGNAN, node task
    mod = gpu.build_module(g)
    data = gpu.device_inputs(g)
    with torch.no_grad():
        y_eval = mod.forward(data)
    mod.dropout = 0.3
    for f in mod.fs:
        for layer in f:
            if isinstance(layer, torch.nn.Dropout):
                layer.p = 0.3
    mod.train()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        torch.manual_seed(7)
        y1 = mod.forward(data)
        torch.manual_seed(7)
        y2 = mod.forward(data)
    assert any("Dropout" in str(m.message) for m in w)
    assert torch.equal(y1, y2) and not torch.equal(y1, y_eval)
    y1.pow(2).sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in mod.parameters() if p.grad is not None)
    mod.eval()
    with torch.no_grad():
        assert torch.equal(mod.forward(data), y_eval)


@pytest.mark.parametrize("variant", ["models_tensor_graph", "models_gnan", "batched"])
def test_same_shaped_graphs_back_to_back_are_not_confused(gpu, variant):
    """The reference's loops upload one graph per step and drop it (trainer.py:46 with batch_size=1): the allocator hands
    the next same-sized graph the previous one's addresses.  Every forward must see ITS graph and ITS features."""
    from gnan_amd import batched, models
    rng = np.random.default_rng(0)
    n, F, H = 12, 4, 8
    torch.manual_seed(0)
    if variant == "batched":
        mod = batched.TensorGNAN(F, 2, 2, hidden_channels=H, device="cuda")
    elif variant == "models_gnan":
        mod = models.GNAN(F, 2, num_layers=3, hidden_channels=H, device="cuda")
    else:
        mod = models.TensorGNAN(F, 2, 3, hidden_channels=H, is_graph_task=True, readout_n_layers=0, device="cuda")
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.7)
    mod = mod.to(gpu.DEV).eval()
    sd = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
    seen_ptrs, recycled = set(), 0
    for step in range(6):
        m = n + 3 * step                                              # more edges every step: different hop structure
        ei = np.stack([rng.integers(0, n, m), rng.integers(0, n, m)])
        ei = np.concatenate([ei, ei[::-1]], axis=1)
        nd, norm = O.pre_process_dense(ei, n)
        x = torch.rand(n, F, generator=torch.Generator().manual_seed(step))
        if variant == "batched":
            hops = torch.from_numpy(O.hop_codes_from_dense(nd).astype(np.float32))          # -1 marks unreachable pairs
            batch = torch.zeros(n, dtype=torch.long)
            want = O.batched_tensor_gnan_forward(x.double(), hops.double(), batch, {k: v.double() for k, v in sd.items()})
            xd, dd = x.to(gpu.DEV), hops.to(gpu.DEV)
            recycled += dd.data_ptr() in seen_ptrs
            seen_ptrs.add(dd.data_ptr())
            with torch.no_grad():
                got = mod(xd, dd, batch.to(gpu.DEV)).cpu()
            del xd, dd
        else:
            data = gpu.Bag(x=x.to(gpu.DEV), edge_index=None, node_distances=nd.to(gpu.DEV),
                           normalization_matrix=norm.to(gpu.DEV))
            recycled += data.node_distances.data_ptr() in seen_ptrs
            seen_ptrs.add(data.node_distances.data_ptr())
            p64 = {k: v.double() for k, v in sd.items()}
            if variant == "models_gnan":
                want = O.gnan_forward(x.double(), nd.double(), norm.double(), p64, True)
            else:
                want = O.tensor_gnan_forward_models(x.double(), nd.double(), norm.double(), p64, True, True, 0)
            with torch.no_grad():
                got = mod(data).cpu()
            del data
        assert O.rel_err(got, want) <= 1e-5, f"step {step}"
    assert recycled > 0, "the allocator never recycled an address: the test did not exercise what it is about"


def test_batched_variant_draws_one_dropout_mask_per_pair(gpu):
    """rho of the batched script carries a Dropout (batched_pyg_main.py:126-131) and runs on all N*N distances (:151): one
    mask per pair.  In training mode the build must not share a mask among the pairs of a hop count, and — the Dropout
    sitting in front of the last Linear — the mean over many draws must be the eval-mode output."""
    import warnings
    from gnan_amd import batched
    torch.manual_seed(0)
    F, C, H = 3, 2, 16
    mod = batched.TensorGNAN(F, C, 2, hidden_channels=H, dropout=0.5, device="cuda").to(gpu.DEV)
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.7)
        for f in mod.fs:                                   # S = F for every node, whatever the shape functions' masks do
            f[3].weight.zero_()
            f[3].bias.fill_(1.0)
    dist = torch.full((4, 4), -1.0)
    for i in range(4):
        dist[i, i] = 0.0
    dist[0, 1] = dist[2, 3] = 1.0                           # rows 0 and 2 list the same hop counts
    x = torch.rand(4, F).to(gpu.DEV)
    dd, batch = dist.to(gpu.DEV), torch.zeros(4, dtype=torch.long, device=gpu.DEV)
    mod.is_graph_task = False
    mod.eval()
    with torch.no_grad():
        want = mod(x, dd, batch)
    assert torch.equal(want[0], want[2])
    mod.train()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            draws = torch.stack([mod(x, dd, batch) for _ in range(600)])
    assert float((draws[:, 0] - draws[:, 2]).abs().max()) > 0          # a shared mask would make rows 0 and 2 equal
    se = draws.std(0) / 600 ** 0.5
    assert bool(((draws.mean(0) - want).abs() <= 5 * se + 1e-6).all())
    y = mod(x, dd, batch)                                               # and it is differentiable
    y.sum().backward()
    assert mod.rho[0].weight.grad is not None and float(mod.rho[0].weight.grad.abs().max()) > 0


@pytest.mark.parametrize("C,rho_per_feature,L", [(1, False, 3), (2, True, 3), (4, False, 2), (3, True, 3)])
def test_reference_order_training_uses_the_sum_first_backward(gpu, C, rho_per_feature, L, monkeypatch):
    """models.py:373-376 under autograd on a CSR graph through the table path: one node whose backward pass is the sum-first
    order's (the fused read-out makes every feature's row gradient the same [N, C] vector).  Outputs and every parameter
    gradient against float64 oracle autograd; C = 3 is not a read-out width the kernel fuses and takes the composed nodes."""
    from gnan_amd import HopGraph, _lib, functional, models
    monkeypatch.setattr(functional, "FMLP_ALGO", _lib.FMLP_PWL)
    used = {"n": 0}
    real = aggregate._ReferenceOrderAggregate.forward

    def counting(ctx, *a):
        used["n"] += 1
        return real(ctx, *a)
    monkeypatch.setattr(aggregate._ReferenceOrderAggregate, "forward", staticmethod(counting))
    rng = np.random.default_rng(C)
    n, F = 3000, 9
    ei = rng.integers(0, n, (2, 4 * n))
    ei = np.unique(np.concatenate([ei, ei[::-1]], axis=1), axis=1)
    ei = ei[:, ei[0] != ei[1]]
    hops = np.full((n, n), -1, dtype=np.int64)
    hops[ei[0], ei[1]] = 1
    np.fill_diagonal(hops, 0)
    rowptr, col, code = O.csr_from_hops(hops, 1)
    g = HopGraph.from_csr(torch.from_numpy(rowptr).to(gpu.DEV), torch.from_numpy(col).to(gpu.DEV), torch.from_numpy(code).to(gpu.DEV),
                          n_cols=n, n_codes=3)
    x = torch.from_numpy(rng.random((n, F), dtype=np.float32))
    x[:, :2] = (x[:, :2] > 0.5).float()
    x[:, -1] = 1.0
    torch.manual_seed(C)
    mod = models.TensorGNAN(F, C, L, hidden_channels=16, rho_per_feature=rho_per_feature, device=gpu.DEV)
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else (2.0 / sum(p.shape)) ** 0.5))
    sd64 = {k: v.detach().double().clone() for k, v in mod.state_dict().items()}
    mod = mod.to(gpu.DEV).eval()
    mod.aggregation_order = "reference"
    data = gpu.Bag(x=x.to(gpu.DEV), edge_index=None, gnan_graph=g)
    target = torch.randn(n, C, generator=gen, dtype=torch.float64)
    y = mod.forward(data)
    ((y - target.to(gpu.DEV).float()) ** 2).sum().backward()
    assert used["n"] == (1 if C in (1, 2, 4) else 0)
    cnt_np = g.cnt.cpu().long().numpy()

    def chain(p, dtype):                                   # shape functions, rho on the distinct distances, shell-form aggregation
        S = O.feature_mlps(x.to(dtype), p).sum(1)
        wt = O.weight_table(O.rho_lut(p, 3, dtype=dtype), cnt_np).expand(n, -1, -1)
        return O.spmm_csr(rowptr, col, code, S, wt)
    with torch.no_grad():
        truth = chain(sd64, torch.float64)
    assert O.rel_err(y.detach().cpu(), truth) <= 1e-5
    g64 = oracle_grads(lambda p: ((chain(p, torch.float64) - target) ** 2).sum(), sd64, torch.float64)
    g32 = oracle_grads(lambda p: ((chain(p, torch.float32) - target.float()) ** 2).sum(), sd64, torch.float32)
    ok, e_build, e_ref, where = grad_rule(module_grads(mod), g64, g32)             # SURVEY 8c on the gradient: max(1e-5, fp32 oracle's own)
    assert ok, f"{where}: build {e_build:.3e} vs fp32 oracle {e_ref:.3e}"


@pytest.mark.parametrize("graph_task", [True, False])
@pytest.mark.parametrize("bias", [True, False])
def test_batched_backward_in_two_launches(gpu, monkeypatch, graph_task, bias):
    """The backward of a small batch (one output channel, as the binary graph tasks have): gnan_small_batch_bwd — every graph's
    share of the gradients of f and rho in one launch (blockIdx.y = graph), their sum in graph order in a second — == float64
    autograd through the oracle of batched_pyg_main.py:133-184 == the general kernels on the blocks' CSR; bit-reproducible;
    graphs of 1 to 128 nodes, per-graph read-out and node outputs."""
    from gnan_amd import batched
    rng = np.random.default_rng(7)
    F, C, H = 5, 1, 16
    sizes = [3, 64, 17, 100, 1, 30, 128, 45, 9, 65] + [int(v) for v in rng.integers(4, 60, 22)]
    batch = []
    for n in sizes:
        hops = rng.integers(-1, 9, (n, n)).astype(np.float32)
        hops[np.arange(n), np.arange(n)] = 0
        batch.append((torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(gpu.DEV),
                      torch.from_numpy(hops).to(gpu.DEV), torch.tensor([int(rng.integers(0, 2))], device=gpu.DEV)))
    x, blocks, y, bv = batched.collate(batch)
    N = sum(sizes)
    dense = torch.full((N, N), -1.0, device=gpu.DEV)
    o = 0
    for _, d, _ in batch:
        dense[o:o + d.shape[0], o:o + d.shape[0]] = d
        o += d.shape[0]
    torch.manual_seed(0)
    mod = batched.TensorGNAN(F, C, 2, hidden_channels=H, device="cuda", bias=bias, is_graph_task=graph_task).to(gpu.DEV).eval()
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    up = torch.randn(len(sizes) if graph_task else N, C, generator=torch.Generator().manual_seed(1)).to(gpu.DEV)
    monkeypatch.setattr(batched, "BATCH_KERNEL_MAX_TOTAL_NODES", 1 << 30)
    calls = []
    real = batched._lib.lib().gnan_small_batch_bwd
    grads = {}
    for tag, fused in (("two_launches", True), ("again", True), ("general", False)):
        monkeypatch.setattr(batched, "BATCH_BACKWARD_KERNEL", fused)
        mod.zero_grad(set_to_none=True)
        out = mod(x, blocks, bv)
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            (out * up).sum().backward()
            torch.cuda.synchronize()
        names = [e.key for e in prof.key_averages()]
        assert any("small_graph_batch_bwd_kernel" in k for k in names) == fused, names
        grads[tag] = {k: p.grad.clone() for k, p in mod.named_parameters()}
    sd = mod.state_dict()
    g64 = oracle_grads(lambda p: (O.batched_tensor_gnan_forward(x.cpu().double(), dense.cpu().double(), bv.cpu(), p, graph_task)
                                  * up.cpu().double()).sum(), sd, torch.float64)
    g32 = oracle_grads(lambda p: (O.batched_tensor_gnan_forward(x.cpu(), dense.cpu(), bv.cpu(), p, graph_task) * up.cpu()).sum(),
                       sd, torch.float32)
    for tag in ("two_launches", "general"):
        ok, e_build, e_ref, where = grad_rule(grads[tag], g64, g32)                # SURVEY 8c on the gradient
        assert ok, f"{tag} {where}: build {e_build:.3e} vs fp32 oracle {e_ref:.3e}"
    for k in g64:
        assert torch.equal(grads["two_launches"][k], grads["again"][k]), k


def test_batched_training_step_replayed_over_slots(gpu):
    """batched.GraphedBatchStep: the training step of batched_pyg_main.py:205-226 (forward, cross-entropy, backward, Adam)
    captured ONCE over slots and replayed for batches of other shapes == the same steps issued eagerly on a twin model: losses
    and parameters after eight batches; a batch that does not fit the slots is refused; the evaluation pass replays too."""
    import copy
    from gnan_amd import batched
    rng = np.random.default_rng(11)
    F, C, H, G = 6, 8, 16, 12
    data = []
    for _ in range(G * 9):
        n = int(rng.integers(3, 70))
        hops = rng.integers(-1, 8, (n, n)).astype(np.float32)
        hops[np.arange(n), np.arange(n)] = 0
        data.append((torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(gpu.DEV),
                     torch.from_numpy(hops).to(gpu.DEV), torch.tensor([int(rng.integers(0, C))], device=gpu.DEV)))
    batches = [batched.collate(data[i:i + G]) for i in range(0, len(data), G)]
    torch.manual_seed(0)
    a = batched.TensorGNAN(F, C, 2, hidden_channels=H, device="cuda").to(gpu.DEV)
    with torch.no_grad():
        for _, p in a.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    b = copy.deepcopy(a)
    loss_fn = torch.nn.CrossEntropyLoss()
    opt_a, opt_b = torch.optim.Adam(a.parameters(), lr=1e-2), torch.optim.Adam(b.parameters(), lr=1e-2)
    eager = []
    for x, blocks, y, bv in batches[:8]:
        opt_a.zero_grad(set_to_none=True)
        loss = loss_fn(a(x, blocks, bv), y)
        loss.backward()
        opt_a.step()
        eager.append(float(loss.detach()))
    x0, b0, y0, _ = batches[0]
    gs = batched.GraphedBatchStep(b, opt_b, loss_fn, x0, b0, y0)   # (runs the first batch's step; a loss MODULE: the fused loss launch)
    replayed = []
    for x, blocks, y, bv in batches[1:8]:
        got = gs.run(x, blocks, y)
        assert got is not None
        replayed.append(float(got[1]))
    assert gs.kernel_nodes <= 8, gs.kernel_nodes
    for e, r in zip(eager[1:], replayed):
        assert abs(e - r) <= TWO_FLOORS * max(abs(e), 1.0), (eager, replayed)          # two routes of the same steps
    scale = max(float(p.detach().abs().max()) for p in a.parameters())
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert float((pa - pb).abs().max()) <= TWO_FLOORS * scale
    # a batch of another number of graphs, or with more nodes than the slots hold, is refused (the caller steps eagerly)
    fewer = batched.collate(data[:G - 1])
    assert gs.run(fewer[0], fewer[1], fewer[2]) is None
    small = batched.GraphedBatchStep(b, None, lambda out, lab: loss_fn(out, lab), x0, b0, y0, node_capacity=int(x0.shape[0]))
    big = max(batches, key=lambda t: t[0].shape[0])
    if big[0].shape[0] > x0.shape[0]:
        assert small.run(big[0], big[1], big[2]) is None
    # evaluation pass replayed over the same kind of slots == the eager forward
    ev = batched.GraphedBatchStep(b, None, lambda out, lab: loss_fn(out, lab), x0, b0, y0)
    b.eval()
    ev2 = batched.GraphedBatchStep(b, None, lambda out, lab: loss_fn(out, lab), x0, b0, y0)
    x, blocks, y, bv = batches[8]
    out, loss, _ = ev2.run(x, blocks, y)
    with torch.no_grad():
        want = b(x, blocks, bv)
    assert float((out - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert abs(float(loss) - float(loss_fn(want, y))) <= 1e-6 * max(1.0, abs(float(loss)))
    del ev


def test_batched_train_epoch_matches_the_scripts_loop(gpu, monkeypatch):
    """batched.train_epoch == the loop of batched_pyg_main.py:205-226 written out (mean loss, mean accuracy, parameters after
    the epoch), with batches that fit replayed from a captured step and a last, smaller batch stepped eagerly; a second epoch
    reuses the captured steps."""
    import copy
    from gnan_amd import batched
    rng = np.random.default_rng(5)
    F, C, H, G = 6, 8, 16, 10
    data = []
    for _ in range(G * 5 + 3):                                            # a last batch of 3 graphs
        n = int(rng.integers(3, 60))
        hops = rng.integers(-1, 6, (n, n)).astype(np.float32)
        hops[np.arange(n), np.arange(n)] = 0
        data.append((torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(gpu.DEV),
                     torch.from_numpy(hops).to(gpu.DEV), torch.tensor([int(rng.integers(0, C))], device=gpu.DEV)))
    batches = [batched.collate(data[i:i + G]) for i in range(0, len(data), G)]
    torch.manual_seed(0)
    a = batched.TensorGNAN(F, C, 2, hidden_channels=H, device="cuda").to(gpu.DEV)
    with torch.no_grad():
        for _, p in a.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    b = copy.deepcopy(a)
    loss_fn = torch.nn.CrossEntropyLoss()
    opt_a, opt_b = torch.optim.Adam(a.parameters(), lr=1e-2), torch.optim.Adam(b.parameters(), lr=1e-2)
    steps = None
    for epoch in range(2):
        tl = ta = 0.0
        for x, blocks, y, bv in batches:                                  # the script's loop
            opt_a.zero_grad(set_to_none=True)
            out = a(x, blocks, bv)
            loss = loss_fn(out, y)
            acc = (out.argmax(dim=-1) == y).float().mean()
            loss.backward()
            opt_a.step()
            tl += float(loss.detach())
            ta += float(acc)
        got_l, got_a, steps = batched.train_epoch(b, batches, loss_fn, opt_b, steps)
        assert abs(got_l - tl / len(batches)) <= TWO_FLOORS * max(1.0, abs(tl / len(batches))), (epoch, got_l, tl / len(batches))
        assert abs(got_a - ta / len(batches)) <= 1e-6 + 1.0 / (G * len(batches)), (epoch, got_a, ta / len(batches))
    assert isinstance(steps.get(G), batched.GraphedBatchStep) and steps[G].step.graph.replays >= 2 * 5 - 1
    assert not isinstance(steps.get(3), bool) or steps.get(3) is not False      # (3 graphs: a step of its own, or eager)
    scale = max(float(p.detach().abs().max()) for p in a.parameters())
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert float((pa - pb).abs().max()) <= TWO_FLOORS * scale


@pytest.mark.parametrize("name", ["case_002_standalone_tensor_node", "case_011_models_tensor_node", "case_026_models_gnan"])
def test_integration_stub_f_sums_on_the_gpu(gpu, name):
    """the ctypes stub a maintainer of the reference would paste into GNAN.py — its ``f_sums`` called
    on the GPU with a PLAIN torch module list carrying a golden's weights (no gnan_amd class involved) == the oracle's
    shape functions summed over the features (GNAN.py:57-62 + :157), float64 truth, 1e-5."""
    from test_abi import integration_stub
    g = Golden(name)
    m = g.meta
    ns = {}
    exec(integration_stub(), ns)
    F = g.inputs["x"].shape[1]

    class Plain(torch.nn.Module):                        # what GNAN.py:24-34 builds
        def __init__(self):
            super().__init__()
            self.out_channels, self.hidden_channels = m["C"], m["H"]
            self.fs = torch.nn.ModuleList(torch.nn.Sequential(
                torch.nn.Linear(1, m["H"]), torch.nn.ReLU(), torch.nn.Dropout(0.0), torch.nn.Linear(m["H"], m["H"]), torch.nn.ReLU(),
                torch.nn.Dropout(0.0), torch.nn.Linear(m["H"], m["C"])) for _ in range(F))
    plain = Plain()
    fs_sd = {k: torch.from_numpy(np.array(v)) for k, v in g.sd.items() if k.startswith("fs.")}
    plain.load_state_dict(fs_sd, strict=True)
    plain = plain.to(gpu.DEV)
    x = torch.from_numpy(np.array(g.inputs["x"]))
    with torch.no_grad():
        got = ns["f_sums"](plain, x.to(gpu.DEV).contiguous())
    want = O.feature_mlps(x.double(), {k: v.double() for k, v in fs_sd.items()}).sum(1)
    assert got.shape == want.shape and O.rel_err(got.cpu(), want) <= 1e-5


def test_batched_train_epoch_steps_an_uncapturable_batch_once(gpu):
    """A loss callable that reads the device (``.item()``) cannot be captured.  Constructing the captured step has by then
    already STEPPED the batch eagerly (its warm-up step is a real one): ``train_epoch`` must take that step as the batch's
    and not step it a second time; later batches run the eager loop; a real error (not a capture failure) is not swallowed."""
    import copy
    import warnings
    from gnan_amd import batched
    rng = np.random.default_rng(6)
    F, C, H, G = 5, 4, 16, 6
    data = []
    for _ in range(G * 3):
        n = int(rng.integers(3, 40))
        hops = rng.integers(-1, 5, (n, n)).astype(np.float32)
        hops[np.arange(n), np.arange(n)] = 0
        data.append((torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(gpu.DEV),
                     torch.from_numpy(hops).to(gpu.DEV), torch.tensor([int(rng.integers(0, C))], device=gpu.DEV)))
    batches = [batched.collate(data[i:i + G]) for i in range(0, len(data), G)]
    torch.manual_seed(0)
    a = batched.TensorGNAN(F, C, 2, hidden_channels=H, device="cuda").to(gpu.DEV)
    with torch.no_grad():
        for _, p in a.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    b = copy.deepcopy(a)
    ce = torch.nn.CrossEntropyLoss()
    seen = []

    def syncing_loss(out, lab):
        loss = ce(out, lab)
        seen.append(loss.item())                                          # a host synchronisation: not capturable
        return loss
    opt_a, opt_b = torch.optim.Adam(a.parameters(), lr=1e-2), torch.optim.Adam(b.parameters(), lr=1e-2)
    for x, blocks, y, bv in batches:                                      # the script's loop: ONE step per batch
        opt_a.zero_grad(set_to_none=True)
        ce(a(x, blocks, bv), y).backward()
        opt_a.step()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        _, _, steps = batched.train_epoch(b, batches, syncing_loss, opt_b)
    assert steps[G] is False and any("eager loop" in str(w.message) for w in caught)
    scale = max(float(p.detach().abs().max()) for p in a.parameters())
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert float((pa - pb).abs().max()) <= 1e-5 * scale               # (a second step of batch 0 moves them by ~1e-2)
    assert all(int(st["step"]) == len(batches) for st in opt_b.state.values())

    def broken_loss(out, lab):
        raise ZeroDivisionError("not a capture failure")
    with pytest.raises(ZeroDivisionError):
        batched.train_epoch(b, batches, broken_loss, opt_b)


def test_batched_replayed_steps_flag_labels_the_fused_loss_does_not_cover(gpu):
    """A replayed batch's labels are never seen by the host.  A class label outside [0, C) — torch's ignore_index rows, left out
    of its mean — is averaged over by the fused cross entropy: the kernel raises a device flag (gnan_loss_args.label_flag, ABI 42)
    and the epoch that read it refuses its own numbers instead of returning another loss than the script's."""
    from gnan_amd import _lib, batched
    rng = np.random.default_rng(9)
    F, C, H, G = 5, 4, 16, 6
    data = []
    for i in range(G * 4):
        n = int(rng.integers(3, 40))
        hops = rng.integers(-1, 5, (n, n)).astype(np.float32)
        hops[np.arange(n), np.arange(n)] = 0
        label = -100 if i == G * 2 + 1 else int(rng.integers(0, C))          # one ignored label, in the third batch
        data.append((torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(gpu.DEV),
                     torch.from_numpy(hops).to(gpu.DEV), torch.tensor([label], device=gpu.DEV)))
    batches = [batched.collate(data[i:i + G]) for i in range(0, len(data), G)]
    torch.manual_seed(0)
    m = batched.TensorGNAN(F, C, 2, hidden_channels=H, device="cuda").to(gpu.DEV)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    loss_fn = torch.nn.CrossEntropyLoss()
    _, _, steps = batched.train_epoch(m, batches[:2], loss_fn, opt)          # clean batches: fine, and the step is captured
    assert isinstance(steps.get(G), batched.GraphedBatchStep)
    with pytest.raises(_lib.GnanHipError, match="outside"):
        batched.train_epoch(m, batches, loss_fn, opt, steps)
    _, _, steps = batched.train_epoch(m, batches[:2], loss_fn, opt, steps)   # the flag was reset: clean batches pass again


def test_batched_batch_with_an_empty_graph_takes_the_csr_route(gpu):
    """A gap in ``batch_vector``'s graph ids is a graph without nodes: the reference's ``scatter_add`` gives it a zero row
    (batched_pyg_main.py:173-181).  The one-launch kernels have no workgroup that would write that row: such a batch must
    take the CSR route, output and gradients as the oracle's."""
    from gnan_amd import batched
    rng = np.random.default_rng(8)
    F, C, H = 4, 3, 8
    sizes = [5, 0, 7, 3]
    N = sum(sizes)
    x = torch.from_numpy(rng.standard_normal((N, F)).astype(np.float32)).to(gpu.DEV)
    bv = torch.tensor([0] * 5 + [2] * 7 + [3] * 3, device=gpu.DEV)
    dense = torch.full((N, N), -1.0, device=gpu.DEV)
    o = 0
    for n in sizes:
        if n:
            hops = rng.integers(-1, 4, (n, n)).astype(np.float32)
            hops[np.arange(n), np.arange(n)] = 0
            dense[o:o + n, o:o + n] = torch.from_numpy(hops).to(gpu.DEV)
        o += n
    torch.manual_seed(0)
    mod = batched.TensorGNAN(F, C, 2, hidden_channels=H, device="cuda").to(gpu.DEV).eval()
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    blocks = mod._blocks(dense, bv)
    assert blocks is not None and blocks.min_nodes == 0 and blocks.n_graphs == 4
    up = torch.randn(4, C, generator=torch.Generator().manual_seed(1)).to(gpu.DEV)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        out = mod(x, dense, bv)
        (out * up).sum().backward()
        torch.cuda.synchronize()
    assert not any("small_graph_batch" in e.key for e in prof.key_averages())
    assert out.shape == (4, C) and float(out[1].abs().max()) == 0.0
    sd = mod.state_dict()
    with torch.no_grad():
        truth = O.batched_tensor_gnan_forward(x.cpu().double(), dense.cpu().double(), bv.cpu(), {k: v.cpu().double() for k, v in sd.items()}, True)
    assert O.rel_err(out.detach().cpu(), truth) <= 1e-5
    g64 = oracle_grads(lambda p: (O.batched_tensor_gnan_forward(x.cpu().double(), dense.cpu().double(), bv.cpu(), p, True)
                                  * up.cpu().double()).sum(), sd, torch.float64)
    g32 = oracle_grads(lambda p: (O.batched_tensor_gnan_forward(x.cpu(), dense.cpu(), bv.cpu(), p, True) * up.cpu()).sum(), sd, torch.float32)
    ok, e_build, e_ref, where = grad_rule(module_grads(mod), g64, g32)
    assert ok, f"{where}: build {e_build:.3e} vs fp32 oracle {e_ref:.3e}"


def test_batched_graphs_in_one_launch(gpu, monkeypatch):
    """f-2 as a path: per-graph hop matrices -> packed code blocks on the device (no (sum N)^2 matrix), all graphs of a batch
    forwarded by ONE launch with the per-graph read-out in its epilogue; == the hop-coded CSR through the general kernels ==
    the float64 oracle of batched_pyg_main.py:133-184; the reference's dense dist_batch gives the same blocks; a matrix
    that is not block-diagonal keeps the general route."""
    from gnan_amd import batched
    rng = np.random.default_rng(3)
    F, C, H = 6, 8, 16
    sizes = [3, 64, 17, 100, 1, 30, 128, 45, 9, 65] + [int(v) for v in rng.integers(4, 60, 30)]
    batch = []
    for n in sizes:
        hops = rng.integers(-1, 7, (n, n)).astype(np.float32)            # -1: unreachable inside the graph
        hops[np.arange(n), np.arange(n)] = 0
        batch.append((torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(gpu.DEV),
                      torch.from_numpy(hops).to(gpu.DEV), torch.tensor([int(rng.integers(0, 2))], device=gpu.DEV)))
    x, blocks, y, bv = batched.collate(batch)
    N = sum(sizes)
    assert x.shape == (N, F) and blocks.n_graphs == len(sizes) and blocks.max_nodes == 128 and bv.shape == (N,)
    dense = torch.full((N, N), -1.0, device=gpu.DEV)                       # what batched_pyg_main.py:69-76 builds
    o = 0
    for _, d, _ in batch:
        dense[o:o + d.shape[0], o:o + d.shape[0]] = d
        o += d.shape[0]
    from_dense = batched.HopBlocks.from_dense(dense, bv)
    assert torch.equal(from_dense.code, blocks.code) and torch.equal(from_dense.node_off, blocks.node_off)
    assert torch.equal(from_dense.code_off, blocks.code_off) and from_dense.n_codes == blocks.n_codes
    want_csr, got_csr = batched.hop_graph_from_counts(dense), blocks.csr()
    for name in ("rowptr", "col", "code"):
        assert torch.equal(getattr(want_csr, name).long(), getattr(got_csr, name).long()), name
    torch.manual_seed(0)
    mod = batched.TensorGNAN(F, C, 2, hidden_channels=H, device="cuda").to(gpu.DEV).eval()
    with torch.no_grad():
        for _, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    outs, grads = {}, {}
    up = torch.randn(len(sizes), C, generator=torch.Generator().manual_seed(1)).to(gpu.DEV)
    monkeypatch.setattr(batched, "BATCH_KERNEL_MAX_TOTAL_NODES", 1 << 30)        # (the policy sends batches this large to the CSR route)
    for tag, kernel, dist in (("launch", True, blocks), ("csr", False, blocks), ("dense", True, dense)):
        monkeypatch.setattr(batched, "BATCH_KERNEL", kernel)
        mod.zero_grad(set_to_none=True)
        out = mod(x, dist, bv)
        (out * up).sum().backward()
        outs[tag] = out.detach()
        grads[tag] = {k: p.grad.clone() for k, p in mod.named_parameters()}
    p64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in mod.state_dict().items()}
    truth = O.batched_tensor_gnan_forward(x.cpu().double(), dense.cpu().double(), bv.cpu(), p64, True)
    (truth * up.cpu().double()).sum().backward()
    assert outs["launch"].shape == (len(sizes), C)
    for tag in ("launch", "csr", "dense"):
        assert O.rel_err(outs[tag].cpu(), truth.detach()) <= 1e-5, tag
    assert torch.equal(outs["launch"], outs["dense"])
    g32 = oracle_grads(lambda p: (O.batched_tensor_gnan_forward(x.cpu(), dense.cpu(), bv.cpu(), p, True) * up.cpu()).sum(),
                       mod.state_dict(), torch.float32)
    for tag in ("launch", "csr"):
        ok, e_build, e_ref, where = grad_rule(grads[tag], {k: v.grad for k, v in p64.items()}, g32)      # SURVEY 8c on the gradient
        assert ok, f"{tag} {where}: build {e_build:.3e} vs fp32 oracle {e_ref:.3e}"
    # node-level outputs through the same launch
    mod.is_graph_task = False
    monkeypatch.setattr(batched, "BATCH_KERNEL", True)
    with torch.no_grad():
        nodes = mod(x, blocks, bv)
    truth_nodes = O.batched_tensor_gnan_forward(x.cpu().double(), dense.cpu().double(), bv.cpu(), {k: v.detach() for k, v in p64.items()}, False)
    assert O.rel_err(nodes.cpu(), truth_nodes) <= 1e-5
    # a listed pair across two graphs: not block-diagonal — the general CSR of the whole matrix, same semantics
    mod.is_graph_task = True
    cross = dense.clone()
    cross[0, 5] = 2.0
    with torch.no_grad():
        got = mod(x, cross, bv)
    want = O.batched_tensor_gnan_forward(x.cpu().double(), cross.cpu().double(), bv.cpu(), {k: v.detach() for k, v in p64.items()}, True)
    assert O.rel_err(got.cpu(), want) <= 1e-5
    with pytest.raises(Exception):
        batched.HopBlocks.from_dense(cross, bv)
    bad = [(batch[0][0], torch.full((3, 3), 0.5, device=gpu.DEV), batch[0][2])]
    with pytest.raises(Exception, match="integer hop counts"):
        batched.collate(bad)
