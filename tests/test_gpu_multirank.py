"""Multi-rank forward and backward with the REAL kernels: 2 and 4 fresh child processes share cuda:0, collectives over gloo.

What the CPU/gloo tests (tests/test_distributed_gloo.py) cannot cover — they substitute the oracle for the kernels — and
what a 1-GPU box can: the product's partitions (all-gather of the operand, halo recompute, halo exchange, feature columns)
driving the HIP look-up / aggregation / backward kernels on every rank, against (i) the single-process HIP result and
(ii) float64 oracle autograd on the whole graph.  Times mean nothing here (the ranks share one device)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import grad_rule, module_grads, oracle_grads
from oracle import gnan_oracle as O

pytestmark = pytest.mark.gpu
N, F_RAW, H, L = 3000, 19, 16, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(C):
    """Same on every rank and in the parent: CPU generators only."""
    from gnan_amd import synthetic as syn
    src, dst = syn.rmat_edges(12, N, 8 * N, seed=0, device="cpu", chunk=1 << 14)
    x = syn.block_features(N, F_RAW + 1, 0, N, seed=1, device="cpu", block=256)
    x[:, :4] = (x[:, :4] > 0.6).float()                         # a few one-hot style columns: exact zeros and ones
    return src, dst, x


def _model(C, dev, algo):
    from gnan_amd import _lib, functional
    from gnan_amd.models import TensorGNAN
    functional.FMLP_ALGO = _lib.FMLP_PWL if algo == "pwl" else _lib.FMLP_AUTO
    torch.manual_seed(3)
    m = TensorGNAN(F_RAW + 1, C, L, hidden_channels=H, rho_per_feature=False, device=dev)
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=gen) * (2.0 / sum(p.shape)) ** 0.5)
            elif not name.startswith("fs.0."):                   # feature 0 keeps the reference's zero biases (kinks at x = 0)
                p.copy_(torch.randn(p.shape, generator=gen) * 0.5)
    return m.to(dev)


def _target(C):
    return torch.sin(torch.arange(N * C, dtype=torch.float32)).view(N, C)


def _worker(rank, world, port, variant, order, C, algo, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gnan_amd  # noqa: F401
        from gnan_amd import synthetic as syn
        from gnan_amd.distributed import (FeaturePartition, VertexPartition, build_exchange_plan, build_halo_plan,
                                          feature_parallel_forward, halo_exchange_forward, halo_recompute_forward,
                                          partitioned_forward, slice_features)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        src, dst, x = _problem(C)
        src, dst, x = src.to(dev), dst.to(dev), x.to(dev)
        m = _model(C, dev, algo).eval()
        part = VertexPartition(N, world, rank)
        stacked = m._stacked("fs", m.fs)                          # proxy leaves: gradients land on the Parameters
        rows = slice(part.lo, part.hi)
        if variant == "vertex":
            g = syn.hop1_csr(src, dst, N, part.lo, part.hi)
            y = partitioned_forward(x[rows], g, stacked, m._lut_global(g), True, part, order=order, out_channels=C)
        elif variant == "halo":
            plan = build_halo_plan(syn.hop1_csr(src, dst, N, part.lo, part.hi), part)
            y = halo_recompute_forward(x[plan.node_ids()].contiguous(), plan, stacked, m._lut_global(plan.graph), True,
                                       order=order, out_channels=C)
        elif variant == "exchange":
            xplan = build_exchange_plan(syn.hop1_csr(src, dst, N, part.lo, part.hi), part)
            y = halo_exchange_forward(x[rows], xplan, stacked, m._lut_global(xplan.halo.graph), True, order=order,
                                      out_channels=C)
        else:                                                     # feature columns; every rank holds the whole output
            fpart = FeaturePartition(F_RAW + 1, world, rank)
            g = syn.hop1_csr(src, dst, N)
            y = feature_parallel_forward(x[:, fpart.lo:fpart.hi].contiguous(), g, slice_features(stacked, fpart.lo, fpart.hi),
                                         m._lut_global(g), True, fpart, out_channels=C)
            rows = slice(0, N)
        loss = ((y - _target(C).to(dev)[rows]) ** 2).sum()
        loss.backward()
        np.save(os.path.join(out_dir, f"y{rank}.npy"), y.detach().cpu().numpy())
        np.savez(os.path.join(out_dir, f"g{rank}.npz"),
                 **{k: (p.grad if p.grad is not None else torch.zeros_like(p)).cpu().numpy() for k, p in m.named_parameters()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,variant,order,C,algo", [
    (2, "vertex", "sum_first", 1, "pwl"), (4, "vertex", "reference", 1, "auto"), (2, "vertex", "sum_first", 3, "pwl"),
    (2, "halo", "reference", 1, "pwl"), (4, "halo", "sum_first", 1, "auto"), (4, "halo", "reference", 1, "pwl"),
    (2, "exchange", "sum_first", 1, "pwl"), (4, "exchange", "reference", 1, "pwl"), (4, "exchange", "sum_first", 3, "auto"),
    (2, "feature", "reference", 1, "pwl"), (4, "feature", "reference", 1, "auto"),
])
def test_ranks_with_the_hip_kernels_equal_single_process_and_oracle(world, variant, order, C, algo, tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    import gnan_amd  # noqa: F401
    from gnan_amd import synthetic as syn
    port = _free_port()
    mp.spawn(_worker, args=(world, port, variant, order, C, algo, str(tmp_path)), nprocs=world, join=True)
    ys = [np.load(tmp_path / f"y{r}.npy") for r in range(world)]
    parts = [np.load(tmp_path / f"g{r}.npz") for r in range(world)]
    if variant == "feature":
        for y in ys[1:]:
            assert np.array_equal(y, ys[0])                      # the all-reduced output is the same on every rank
        got = ys[0]
    else:
        got = np.concatenate(ys)
    assert got.shape == (N, C)

    # (i) the same model in this process, one rank, HIP kernels
    dev = torch.device("cuda", 0)
    src, dst, x = _problem(C)
    m = _model(C, dev, algo).eval()
    m.aggregation_order = order

    class Bag:
        pass
    data = Bag()
    data.x, data.edge_index, data.gnan_graph = x.to(dev), None, syn.hop1_csr(src.to(dev), dst.to(dev), N)
    y1 = m.forward(data)
    ((y1 - _target(C).to(dev)) ** 2).sum().backward()
    assert O.rel_err(torch.from_numpy(got), y1.detach().cpu().double()) <= 5e-6   # (the ranks add their column sums in another order)

    # (ii) float64 oracle autograd on the whole graph
    p64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items()}
    g = data.gnan_graph
    S = O.feature_mlps(x.double(), p64).sum(1)
    wt = O.weight_table(O.rho_lut(p64, 3, dtype=torch.float64), g.cnt.cpu().long().numpy()).expand(N, -1, -1)
    truth = O.spmm_csr(g.rowptr.cpu().long().numpy(), g.col.cpu().numpy(), g.code.cpu().numpy(), S, wt)
    # The tolerance rule: max(1e-5, the error of the FLOAT32 REFERENCE against the same float64 truth).
    # The float32 reference is the oracle's dense restatement of the module under test in the reference's own operation order
    # (models.py:358-384: rho on all N x N distances, the N-term matmul per column, then the sum over features) on the dense
    # inputs this K = 1 graph stands for (hops beyond K zeroed, shell sizes recounted: DESIGN.md section 2).  On this problem
    # every feature column carries a rest-bucket term of ~0.12 that cancels against the others' down to outputs of ~0.02:
    # ill-conditioned in float32 for ANY evaluation, the reference's included.
    if order == "reference":
        sd32 = {k: v.detach().cpu().float() for k, v in m.state_dict().items()}
        fx32 = O.feature_mlps(x.float(), sd32)[:, :, 0]                          # [N, F]   models.py:360-365
        lut32 = O.rho_lut(sd32, 3)[:, 0]
        cntf = g.cnt.cpu().float().clamp_min(1)
        rp = g.rowptr.cpu().long()
        rows_of = torch.repeat_interleave(torch.arange(N), rp[1:] - rp[:-1])
        cols, codes = g.col.cpu().long(), g.code.cpu().long()
        # m[i, j] = rho(nd[i, j]) / norm[i, j] for ALL pairs (models.py:368-370): the rest weight everywhere, the listed pairs'
        # weights on top (this synthetic graph lists some pairs twice: each listing counts, as in the CSR)
        w_rest = lut32[2] / cntf[:, 2]
        M = w_rest.unsqueeze(1).expand(N, N).clone()
        M.index_put_((rows_of, cols), lut32[codes] / cntf[rows_of, codes] - w_rest[rows_of], accumulate=True)
        ref32 = torch.matmul(M, fx32).sum(dim=1, keepdim=True)                   # models.py:371-376: N-term dots, then the features
        e_ref = O.rel_err(ref32, truth.detach())
    else:
        e_ref = 0.0                                                             # sum-first (GNAN.py:157) is well-conditioned: 1e-5
    err = O.rel_err(torch.from_numpy(got), truth.detach())
    assert err <= max(1e-5, e_ref), (err, e_ref, order)
    ((truth - _target(C).double()) ** 2).sum().backward()
    # gradients by the same rule: max(1e-5, the float32 oracle's own gap to float64) — the sum-first chain in float32
    cnt_np = g.cnt.cpu().long().numpy()
    rp_np, col_np, code_np = g.rowptr.cpu().long().numpy(), g.col.cpu().numpy(), g.code.cpu().numpy()

    def chain32(p):
        S = O.feature_mlps(x.float(), p).sum(1)
        wt = O.weight_table(O.rho_lut(p, 3), cnt_np).expand(N, -1, -1)
        return ((O.spmm_csr(rp_np, col_np, code_np, S, wt) - _target(C)) ** 2).sum()
    g32 = oracle_grads(chain32, m.state_dict(), torch.float32)
    g64 = {k: v.grad for k, v in p64.items()}
    # feature partition: every rank back-propagates the SAME loss on the whole output into its own columns' shape
    # functions; rho sees the whole loss on every rank (its gradient is the sum over the ranks' partial outputs)
    summed = {k: sum(p[k] for p in parts) for k in p64}          # the data-parallel all-reduce of p.grad, done here
    ok, e_build, e_ref, where = grad_rule(summed, g64, g32)
    assert ok, f"{where}: ranks vs oracle {e_build:.3e} (fp32 oracle {e_ref:.3e})"
    ok, e_build, e_ref, where = grad_rule(module_grads(m), g64, g32)
    assert ok, f"{where}: single process vs oracle {e_build:.3e} (fp32 oracle {e_ref:.3e})"


def _bench_line(extra, timeout=1500):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--scale", "18", "--nodes", "200000", "--edges", "2000000"] + extra
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-6000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("partition", ["halo", "vertex"])
def test_bench_starts_its_own_ranks(partition):
    """``python bench.py --gpus 2`` with no launcher around it (WORLD_SIZE unset) must run TWO ranks — the parent starts
    ``torch.distributed.run`` as a child before touching the GPU — and say so in its line; here both share cuda:0 over gloo.
    The partitioned result equals the one-rank run of the same graph."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    env_had = os.environ.pop("WORLD_SIZE", None)
    try:
        two = _bench_line(["--gpus", "2", "--same-device", "--backend", "gloo", "--partition", partition])
        one = _bench_line(["--gpus", "1"])
    finally:
        if env_had is not None:
            os.environ["WORLD_SIZE"] = env_had
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["config"]["partition"] == f"{partition} x2"
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1
    assert abs(two["checksum"] - one["checksum"]) <= 1e-5 * abs(one["checksum"]), (two["checksum"], one["checksum"])


def test_bench_line_survives_extra_partitions_that_do_not_finish():
    """A collective of the ``alt_partitions`` legs that never returns (first contact with a real 8-GPU node) must not cost the
    measured line: behind ``--alt-timeout`` every rank leaves with 0 and rank 0 prints the line, the legs marked unfinished."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    env_had = os.environ.pop("WORLD_SIZE", None)
    try:
        cut = _bench_line(["--gpus", "2", "--same-device", "--backend", "gloo", "--partition", "halo", "--alt-timeout", "0.001"])
        full = _bench_line(["--gpus", "2", "--same-device", "--backend", "gloo", "--partition", "halo"])
    finally:
        if env_had is not None:
            os.environ["WORLD_SIZE"] = env_had
    assert cut["value"] > 0 and set(cut["alt_partitions"]) == {"vertex", "exchange"}
    assert any("not finished" in str(v.get("error", "")) for v in cut["alt_partitions"].values()), cut["alt_partitions"]
    for name, leg in full["alt_partitions"].items():
        assert "error" not in leg, (name, leg)
        assert abs(leg["checksum"] - full["checksum"]) <= 1e-5 * abs(full["checksum"]), (name, leg["checksum"], full["checksum"])
