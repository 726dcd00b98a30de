"""World-size-2 (and 3, ragged) CPU tests of the vertex-partitioned forward over gloo.

The collective plumbing (partition arithmetic, padding of the last shard, all-gather of the operand, local
rest-bucket totals, global column ids) is exactly what runs over RCCL on the GPUs; the two local compute
steps are substituted by the oracle here (tests may use it, the product never does)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gnan_amd  # noqa: F401
from gnan_amd import synthetic as syn
from gnan_amd.distributed import VertexPartition, partitioned_forward
from oracle import gnan_oracle as O
from test_pwl_tables import mlp_state, stack


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_compute(sd, C):
    def feature_mlps(x, stacked, sum_features, return_total=False, total_rows=None):
        fx = O.feature_mlps(x, sd)
        out = fx.sum(1) if sum_features else fx.reshape(x.shape[0], -1)
        return (out, out[:total_rows].sum(0)) if return_total else out

    def column_sums(S):
        return S.sum(0)

    def aggregate(g, S, lut, use_cnt, s_total=None, reduce_channels=0, **_backward_only):
        rowptr, col, code = g.rowptr.long().numpy(), g.col.numpy(), g.code.numpy()
        cnt = g.cnt.long().numpy() if use_cnt else None
        wt = O.weight_table(lut, cnt).expand(g.n_rows, -1, -1)
        assert torch.allclose(s_total, S.sum(0), rtol=1e-5, atol=1e-5)
        Y = O.spmm_csr(rowptr, col, code, S, wt)
        return Y.view(Y.shape[0], -1, reduce_channels).sum(1) if reduce_channels else Y
    return {"feature_mlps": feature_mlps, "column_sums": column_sums, "aggregate": aggregate}


def _problem(n, F, C):
    src, dst = syn.rmat_edges(9, n, 6 * n, seed=0, device="cpu", chunk=1 << 12)
    x = syn.block_features(n, F, 0, n, seed=1, device="cpu", block=64)
    sd = mlp_state(F, 3, 8, C, True, seed=2)
    sd.update({"rho.0.weight": torch.randn(8, 1, generator=torch.Generator().manual_seed(3)),
               "rho.0.bias": torch.randn(8, generator=torch.Generator().manual_seed(4)),
               "rho.2.weight": torch.randn(1, 8, generator=torch.Generator().manual_seed(5)),
               "rho.2.bias": torch.randn(1, generator=torch.Generator().manual_seed(6))})
    return src, dst, x, sd


def _worker(rank, world, port, n, F, C, order, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src, dst, x, sd = _problem(n, F, C)
        part = VertexPartition(n, world, rank)
        g = syn.hop1_csr(src, dst, n, part.lo, part.hi)
        lut = O.rho_lut(sd, 3)
        y = partitioned_forward(x[part.lo:part.hi], g, stack(sd, F, 3, 8, C, True), lut, True, part, order=order,
                                out_channels=C, compute=_oracle_compute(sd, C))
        np.save(os.path.join(out_dir, f"y{rank}.npy"), y.numpy())
        np.save(os.path.join(out_dir, f"x{rank}.npy"), x[part.lo:part.hi].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,order", [(2, 300, "sum_first"), (2, 300, "reference"), (3, 301, "reference")])
def test_partitioned_forward_equals_single_process(world, n, order, tmp_path):
    F, C = 5, 1
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, F, C, order, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    src, dst, x, sd = _problem(n, F, C)
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(world)]), x.numpy())
    g = syn.hop1_csr(src, dst, n)
    S = O.feature_mlps(x, sd).sum(1)
    wt = O.weight_table(O.rho_lut(sd, 3), g.cnt.long().numpy()).expand(n, -1, -1)
    want = O.spmm_csr(g.rowptr.long().numpy(), g.col.numpy(), g.code.numpy(), S, wt)
    assert got.shape == (n, C)
    assert O.rel_err(torch.from_numpy(got), want.double()) <= 1e-5


def _feature_worker(rank, world, port, n, F, C, out_dir):
    from gnan_amd.distributed import FeaturePartition, feature_parallel_forward, slice_features
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src, dst, x, sd = _problem(n, F, C)
        part = FeaturePartition(F, world, rank)
        g = syn.hop1_csr(src, dst, n)
        lut = O.rho_lut(sd, 3)
        sd_local = {}
        for k in range(part.lo, part.hi):                       # the oracle sees only this rank's shape functions
            for key, v in sd.items():
                if key.startswith(f"fs.{k}."):
                    sd_local[f"fs.{k - part.lo}." + key.split(".", 2)[2]] = v
        st = slice_features(stack(sd, F, 3, 8, C, True), part.lo, part.hi)
        assert st.F == part.hi - part.lo
        y = feature_parallel_forward(x[:, part.lo:part.hi].contiguous(), g, st, lut, True, part, out_channels=C,
                                     compute=_oracle_compute(sd_local, C))
        np.save(os.path.join(out_dir, f"y{rank}.npy"), y.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,F", [(2, 5), (3, 4), (4, 3)])
def test_feature_parallel_forward_equals_single_process(world, F, tmp_path):
    n, C = 200, 1
    port = _free_port()
    mp.spawn(_feature_worker, args=(world, port, n, F, C, str(tmp_path)), nprocs=world, join=True)
    src, dst, x, sd = _problem(n, F, C)
    g = syn.hop1_csr(src, dst, n)
    S = O.feature_mlps(x, sd).sum(1)
    wt = O.weight_table(O.rho_lut(sd, 3), g.cnt.long().numpy()).expand(n, -1, -1)
    want = O.spmm_csr(g.rowptr.long().numpy(), g.col.numpy(), g.code.numpy(), S, wt)
    for r in range(world):                                       # every rank holds the full, identical output
        got = np.load(tmp_path / f"y{r}.npy")
        assert O.rel_err(torch.from_numpy(got), want.double()) <= 1e-5


def _halo_compute(sd, C):
    base = _oracle_compute(sd, C)

    def aggregate(g, S, lut, use_cnt, s_total=None, reduce_channels=0, **_backward_only):
        rowptr, col, code = g.rowptr.long().numpy(), g.col.numpy(), g.code.numpy()
        wt = O.weight_table(lut, g.cnt.long().numpy() if use_cnt else None).expand(g.n_rows, -1, -1)
        # the oracle forms the rest bucket from the operand it is given; the compact operand is not the whole graph
        Y = O.spmm_csr(rowptr, col, code, S, wt) + wt[:, -1] * (s_total - S.sum(0)).unsqueeze(0)
        return Y.view(Y.shape[0], -1, reduce_channels).sum(1) if reduce_channels else Y
    return {**base, "aggregate": aggregate}


def _cut(n, world, rank, src, cut):
    """Equal row counts, or blocks of equal cost (what bench.py cuts the halo / exchange partitions by)."""
    from gnan_amd.distributed import balanced_bounds
    return VertexPartition(n, world, rank, balanced_bounds(torch.bincount(src, minlength=n) + 1, world) if cut == "cost" else None)


def _halo_worker(rank, world, port, n, F, C, order, out_dir, overlap=False, cut="rows"):
    from gnan_amd.distributed import build_halo_plan, halo_recompute_forward
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src, dst, x, sd = _problem(n, F, C)
        part = _cut(n, world, rank, src, cut)
        g = syn.hop1_csr(src, dst, n, part.lo, part.hi)
        plan = build_halo_plan(g, part)
        ids = plan.node_ids()
        # index work is bit-exact: compact ids point at the same global nodes, own rows first, halo ascending
        assert torch.equal(ids[plan.graph.col.long()], g.col.long())
        assert torch.equal(ids[: plan.n_own], torch.arange(part.lo, part.hi))
        halo = ids[plan.n_own:]
        assert bool((halo[1:] > halo[:-1]).all()) and not bool(((halo >= part.lo) & (halo < part.hi)).any())
        assert set(halo.tolist()) == set(g.col.long().tolist()) - set(range(part.lo, part.hi))
        compute = _halo_compute(sd, C)
        if overlap:   # inference path: aggregate against zero column sums while their all-reduce is in flight, add the rest after
            from cpu_kernels import rest_total_term                # (the product's gnan_rest_term_add needs the GPU)
            compute["rest_total_term"] = rest_total_term
        with torch.no_grad() if overlap else torch.enable_grad():
            y = halo_recompute_forward(x[ids], plan, stack(sd, F, 3, 8, C, True), O.rho_lut(sd, 3), True, order=order,
                                       out_channels=C, compute=compute)
        np.save(os.path.join(out_dir, f"y{rank}.npy"), y.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,order,overlap,cut", [(2, 300, "reference", False, "rows"), (3, 301, "reference", False, "rows"),
                                                       (2, 300, "sum_first", False, "rows"), (2, 300, "reference", True, "rows"),
                                                       (3, 301, "sum_first", True, "rows"), (3, 301, "reference", False, "cost"),
                                                       (2, 300, "reference", True, "cost")])
def test_halo_recompute_forward_equals_single_process(world, n, order, overlap, cut, tmp_path):
    F, C = 5, 1
    port = _free_port()
    mp.spawn(_halo_worker, args=(world, port, n, F, C, order, str(tmp_path), overlap, cut), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])   # blocks in rank order, whatever their sizes
    src, dst, x, sd = _problem(n, F, C)
    g = syn.hop1_csr(src, dst, n)
    S = O.feature_mlps(x, sd).sum(1)
    wt = O.weight_table(O.rho_lut(sd, 3), g.cnt.long().numpy()).expand(n, -1, -1)
    want = O.spmm_csr(g.rowptr.long().numpy(), g.col.numpy(), g.code.numpy(), S, wt)
    assert got.shape == (n, C)
    assert O.rel_err(torch.from_numpy(got), want.double()) <= 1e-5


def _grad_problem(n, F):
    src, dst, x, sd = _problem(n, F, 1)
    return src, dst, x, {k: v.double() for k, v in sd.items()}


def _grad_worker(rank, world, port, n, F, partition, order, out_dir):
    """One rank of a multi-rank forward + backward.  The aggregation is the product's own autograd function running on
    the stand-in launchers of tests/cpu_kernels.py, the shape functions are the oracle's (differentiable torch); the
    collectives — and what the backward pass sends through them — are the product's."""
    import cpu_kernels
    from gnan_amd.distributed import (FeaturePartition, build_halo_plan, feature_parallel_forward, halo_recompute_forward,
                                      slice_features)
    from gnan_amd.aggregate import rho_aggregate
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu_kernels.install()
    try:
        src, dst, x, sd = _grad_problem(n, F)
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        lut = O.rho_lut(leaves, 3, dtype=torch.float64).float()

        def features_of(keys_lo, keys_hi):
            def feature_mlps(xx, stacked, sum_features, return_total=False, total_rows=None, **_):
                local = {f"fs.{k - keys_lo}." + key.split(".", 2)[2]: v for key, v in leaves.items()
                         for k in range(keys_lo, keys_hi) if key.startswith(f"fs.{k}.")}
                fx = O.feature_mlps(xx.double(), local).float()
                out = fx.sum(1) if sum_features else fx.reshape(xx.shape[0], -1)
                return (out, out[:total_rows].sum(0).detach()) if return_total else out
            return feature_mlps
        compute = {"feature_mlps": features_of(0, F), "aggregate": rho_aggregate}
        part = VertexPartition(n, world, rank)
        if partition == "halo":
            plan = build_halo_plan(syn.hop1_csr(src, dst, n, part.lo, part.hi), part)
            y = halo_recompute_forward(x[plan.node_ids()], plan, None, lut, True, order=order, out_channels=1,
                                       compute=compute)
            rows = slice(part.lo, part.hi)
        elif partition == "vertex":
            g = syn.hop1_csr(src, dst, n, part.lo, part.hi)
            y = partitioned_forward(x[part.lo:part.hi], g, None, lut, True, part, order=order, out_channels=1,
                                    compute=compute)
            rows = slice(part.lo, part.hi)
        else:
            fpart = FeaturePartition(F, world, rank)
            compute["feature_mlps"] = features_of(fpart.lo, fpart.hi)
            y = feature_parallel_forward(x[:, fpart.lo:fpart.hi].contiguous(), syn.hop1_csr(src, dst, n), None, lut, True,
                                         fpart, out_channels=1, compute=compute)
            rows = slice(0, n)
        target = torch.sin(torch.arange(n, dtype=torch.float32)).view(-1, 1)
        loss = ((y - target[rows]) ** 2).sum()
        # feature partition: every rank evaluates the SAME loss on the whole (summed) output and back-propagates it into
        # its own partial output, so the ranks' gradients add up like the partials did
        loss.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in leaves.items()}
        np.savez(os.path.join(out_dir, f"g{rank}.npz"), **grads)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,partition,order", [(2, 120, "halo", "reference"), (3, 121, "halo", "sum_first"),
                                                     (2, 120, "vertex", "sum_first"), (3, 121, "vertex", "reference"),
                                                     (2, 120, "feature", "reference")])
def test_multi_rank_backward_adds_up_to_the_single_process_gradient(world, n, partition, order, tmp_path):
    """Sum over the ranks of the parameter gradients == the single-process gradient (float64 oracle autograd).  The
    rest-bucket total couples every output row to every operand row of every rank; the backward pass has to carry that
    through the same collectives as the forward."""
    F = 4
    port = _free_port()
    mp.spawn(_grad_worker, args=(world, port, n, F, partition, order, str(tmp_path)), nprocs=world, join=True)
    src, dst, x, sd = _grad_problem(n, F)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    g = syn.hop1_csr(src, dst, n)
    S = O.feature_mlps(x.double(), leaves).sum(1)
    wt = O.weight_table(O.rho_lut(leaves, 3, dtype=torch.float64), g.cnt.long().numpy()).expand(n, -1, -1)
    y = O.spmm_csr(g.rowptr.long().numpy(), g.col.numpy(), g.code.numpy(), S, wt)
    target = torch.sin(torch.arange(n, dtype=torch.float32)).view(-1, 1).double()
    ((y - target) ** 2).sum().backward()
    scale = max(float(v.grad.abs().max()) for v in leaves.values())
    parts = [np.load(tmp_path / f"g{r}.npz") for r in range(world)]
    for k, v in leaves.items():
        got = sum(p[k] for p in parts)
        err = float(np.abs(got - v.grad.numpy()).max()) / scale
        assert err <= 1e-5, f"{k}: {err:.3e}"                     # float64 truth: the floor of SURVEY 8c


def _exchange_worker(rank, world, port, n, F, order, out_dir, with_grad, cut="rows"):
    import cpu_kernels
    from gnan_amd.distributed import build_exchange_plan, halo_exchange_forward
    from gnan_amd.aggregate import rho_aggregate
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu_kernels.install()
    try:
        src, dst, x, sd = _grad_problem(n, F)
        leaves = {k: v.clone().requires_grad_(with_grad) for k, v in sd.items()}
        lut = O.rho_lut(leaves, 3, dtype=torch.float64).float()
        part = _cut(n, world, rank, src, cut)
        g = syn.hop1_csr(src, dst, n, part.lo, part.hi)
        xplan = build_exchange_plan(g, part)
        # index work is bit-exact: what a rank is asked to send is what the asker's halo lists, in the asker's order
        halo = xplan.halo.halo
        assert sum(xplan.recv_counts) == halo.numel() and xplan.recv_counts[rank] == 0
        for idx in xplan.send_rows:
            assert idx is None or (idx.numel() == 0) or (0 <= int(idx.min()) and int(idx.max()) < part.hi - part.lo)

        def feature_mlps(xx, stacked, sum_features, return_total=False, total_rows=None, **_):
            fx = O.feature_mlps(xx.double(), leaves).float()
            out = fx.sum(1) if sum_features else fx.reshape(xx.shape[0], -1)
            return (out, out[:total_rows].sum(0).detach()) if return_total else out
        compute = {"feature_mlps": feature_mlps, "aggregate": rho_aggregate}
        ctx = torch.enable_grad() if with_grad else torch.no_grad()
        with ctx:
            y = halo_exchange_forward(x[part.lo:part.hi], xplan, None, lut, True, order=order, out_channels=1, compute=compute)
            if with_grad:
                target = torch.sin(torch.arange(n, dtype=torch.float32)).view(-1, 1)
                ((y - target[part.lo:part.hi]) ** 2).sum().backward()
                np.savez(os.path.join(out_dir, f"g{rank}.npz"),
                         **{k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in leaves.items()})
        np.save(os.path.join(out_dir, f"y{rank}.npy"), y.detach().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,order,cut", [(2, 120, "sum_first", "rows"), (3, 121, "reference", "rows"), (4, 90, "sum_first", "rows"),
                                               (3, 121, "reference", "cost"), (4, 90, "sum_first", "cost")])
def test_halo_exchange_forward_and_backward_equal_single_process(world, n, order, cut, tmp_path):
    """Only the listed remote operand rows travel (point-to-point all-to-all-v); forward == single process, and the sum
    over the ranks of the parameter gradients == the single-process gradient (the exchange runs in reverse in backward)."""
    F = 4
    port = _free_port()
    mp.spawn(_exchange_worker, args=(world, port, n, F, order, str(tmp_path), True, cut), nprocs=world, join=True)
    src, dst, x, sd = _grad_problem(n, F)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    g = syn.hop1_csr(src, dst, n)
    S = O.feature_mlps(x.double(), leaves).sum(1)
    wt = O.weight_table(O.rho_lut(leaves, 3, dtype=torch.float64), g.cnt.long().numpy()).expand(n, -1, -1)
    y = O.spmm_csr(g.rowptr.long().numpy(), g.col.numpy(), g.code.numpy(), S, wt)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    assert O.rel_err(torch.from_numpy(got), y.detach()) <= 1e-5
    target = torch.sin(torch.arange(n, dtype=torch.float32)).view(-1, 1).double()
    ((y - target) ** 2).sum().backward()
    scale = max(float(v.grad.abs().max()) for v in leaves.values())
    parts = [np.load(tmp_path / f"g{r}.npz") for r in range(world)]
    for k, v in leaves.items():
        err = float(np.abs(sum(p[k] for p in parts) - v.grad.numpy()).max()) / scale
        assert err <= 1e-5, f"{k}: {err:.3e}"                     # float64 truth: the floor of SURVEY 8c


def test_choose_partition_by_exchanged_bytes():
    from gnan_amd.distributed import choose_partition
    assert choose_partition(10_000_000, 64, 1, 8, "reference") == "halo"        # inputs replicated: no exchange
    assert choose_partition(10_000_000, 64, 1, 8, "reference", replicated_inputs=False) == "feature"  # 2.2 GB vs 70 MB
    assert choose_partition(10_000_000, 64, 1, 8, "sum_first") == "vertex"      # narrow operand: gather it
    assert choose_partition(10_000_000, 2, 1, 8, "reference", replicated_inputs=False) == "vertex"
    assert choose_partition(10_000_000, 64, 1, 1, "reference") == "vertex"


def test_partition_arithmetic():
    for n, world in [(10, 3), (7, 8), (16, 4), (1, 2)]:
        parts = [VertexPartition(n, world, r) for r in range(world)]
        assert parts[0].lo == 0 and parts[-1].hi == n
        assert all(a.hi == b.lo for a, b in zip(parts, parts[1:]))
        assert all(p.hi - p.lo <= p.block for p in parts)


def test_blocks_of_equal_cost():
    """distributed.balanced_bounds: contiguous blocks, every node in exactly one, costs within a row of the mean; the partition
    answers ownership queries from the same bounds; the all-gather variant refuses an unequal cut."""
    from gnan_amd.distributed import ROW_COST_IN_PAIRS, HALO_ROWS_PER_PAIR, balanced_bounds
    gen = torch.Generator().manual_seed(0)
    n, world = 5000, 8
    deg = (torch.rand(n, generator=gen) ** -1.2).long().clamp_(max=900) + 1            # heavy-tailed
    bounds = balanced_bounds(deg, world)
    assert len(bounds) == world + 1 and bounds[0] == 0 and bounds[-1] == n and all(a < b for a, b in zip(bounds, bounds[1:]))
    cost = deg.double() * (1 + ROW_COST_IN_PAIRS * HALO_ROWS_PER_PAIR) + ROW_COST_IN_PAIRS
    shares = [float(cost[a:b].sum()) for a, b in zip(bounds, bounds[1:])]
    mean = sum(shares) / world
    assert max(abs(s_ - mean) for s_ in shares) <= float(cost.max()) + 1e-9          # off by at most one row's cost
    rows = [b - a for a, b in zip(bounds, bounds[1:])]
    assert max(rows) > min(rows)                                                      # ... which equal row counts are not
    parts = [VertexPartition(n, world, r, bounds) for r in range(world)]
    assert [(p.lo, p.hi) for p in parts] == list(zip(bounds, bounds[1:])) and not parts[0].uniform
    nodes = torch.arange(n)
    owner = parts[0].owner_of(nodes)
    for r, p in enumerate(parts):
        assert bool((owner[p.lo:p.hi] == r).all())
    assert torch.equal(nodes - parts[0].lo_of(owner), torch.cat([torch.arange(b - a) for a, b in zip(bounds, bounds[1:])]))
    with pytest.raises(ValueError):
        parts[0].block
    with pytest.raises(ValueError):
        VertexPartition(n, world, 0, (0, 10, 5) + bounds[3:])
    assert balanced_bounds(deg, 1) == (0, n)
    assert balanced_bounds(torch.ones(3, dtype=torch.int64), 8)[-1] == 3            # more ranks than nodes: empty blocks allowed
    uni = VertexPartition(10, 3, 1)
    assert uni.uniform and torch.equal(uni.owner_of(torch.arange(10)), torch.arange(10) // 4)
