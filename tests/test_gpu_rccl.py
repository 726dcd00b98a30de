"""First contact with RCCL: a ONE-rank process group on backend ``nccl`` (what a 1-GPU box can hold — RCCL refuses two
ranks on one device) drives every partition of ``gnan_amd.distributed`` with its collectives switched on
(``distributed.ALWAYS_COMMUNICATE``): communicator creation, ``all_gather_into_tensor``, ``reduce_scatter_tensor`` (the
branch gloo never enters), ``all_reduce`` (blocking, ``async_op`` and inside the backward pass), the batched point-to-point
exchange with an empty peer list, and the all-reduce CAPTURED inside ``SharePipeline``'s two hipGraphs, replayed six times.
On a world of one every collective is the identity, so each result must equal the same forward + backward without a group.

The reference has no distributed code (/root/reference/main.py:49-52 picks one device); this is
build-only.  The worker is a fresh child process (a process group and its communicator do not belong in the pytest process).
"""
import json
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N, F_RAW, H, L = 3000, 19, 16, 3

CASES = [("vertex", "sum_first", 1, True), ("vertex", "reference", 1, True), ("vertex", "sum_first", 3, True),
         ("halo", "reference", 1, True), ("halo", "sum_first", 1, True), ("halo", "reference", 1, False),
         ("exchange", "sum_first", 1, True), ("exchange", "reference", 1, True),
         ("feature", "reference", 1, True), ("feature", "reference", 1, False)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, out_path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    report = {"backend": dist.get_backend(), "world": dist.get_world_size(), "cases": [], "collectives": {}}
    try:
        import gnan_amd  # noqa: F401
        from gnan_amd import distributed as D
        from gnan_amd import functional
        from gnan_amd import synthetic as syn
        from gnan_amd.functional import stack_mlps
        from test_gpu_multirank import _model, _problem, _target

        # ---- the collectives themselves, as distributed.py calls them ----------------------------------------------
        a = torch.arange(12, dtype=torch.float32, device=dev).view(4, 3)
        full = torch.empty_like(a)
        dist.all_gather_into_tensor(full, a)
        back = D._reduce_scatter_sum(a.clone(), 4, None)                  # nccl: reduce_scatter_tensor
        s = a.clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        w = a.clone()
        dist.all_reduce(w, op=dist.ReduceOp.MAX, async_op=True).wait()
        D._peer_exchange([None], [None], None)                            # no peers: nothing to post
        dist.barrier()
        torch.cuda.synchronize()
        report["collectives"] = {k: bool(torch.equal(v, a)) for k, v in
                                 {"all_gather_into_tensor": full, "reduce_scatter_tensor": back, "all_reduce": s,
                                  "all_reduce_async": w}.items()}

        # ---- every partition, forward + backward, with and without the group ------------------------------------
        def run(variant, order, C, grad, communicate):
            D.ALWAYS_COMMUNICATE = communicate
            src, dst, x = _problem(C)
            src, dst, x = src.to(dev), dst.to(dev), x.to(dev)
            m = _model(C, dev, "pwl").eval()
            part = D.VertexPartition(N, 1, 0)
            stacked = m._stacked("fs", m.fs)
            with torch.set_grad_enabled(grad):
                if variant == "vertex":
                    g = syn.hop1_csr(src, dst, N)
                    y = D.partitioned_forward(x, g, stacked, m._lut_global(g), True, part, order=order, out_channels=C)
                elif variant == "halo":
                    plan = D.build_halo_plan(syn.hop1_csr(src, dst, N), part)
                    y = D.halo_recompute_forward(x[plan.node_ids()].contiguous(), plan, stacked, m._lut_global(plan.graph),
                                                 True, order=order, out_channels=C)
                elif variant == "exchange":
                    xplan = D.build_exchange_plan(syn.hop1_csr(src, dst, N), part)
                    assert xplan.halo.halo.numel() == 0
                    y = D.halo_exchange_forward(x, xplan, stacked, m._lut_global(xplan.halo.graph), True, order=order,
                                                out_channels=C)
                else:
                    fpart = D.FeaturePartition(F_RAW + 1, 1, 0)
                    g = syn.hop1_csr(src, dst, N)
                    y = D.feature_parallel_forward(x, g, D.slice_features(stacked, 0, F_RAW + 1), m._lut_global(g), True,
                                                   fpart, out_channels=C)
                grads = {}
                if grad:
                    ((y - _target(C).to(dev)) ** 2).sum().backward()
                    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
            D.ALWAYS_COMMUNICATE = False
            return y.detach().clone(), grads

        for variant, order, C, grad in CASES:
            y1, g1 = run(variant, order, C, grad, True)
            y0, g0 = run(variant, order, C, grad, False)
            torch.cuda.synchronize()
            scale = float(y0.abs().max())
            gscale = max([float(v.abs().max()) for v in g0.values()] or [1.0])
            report["cases"].append({
                "case": [variant, order, C, grad], "y": float((y1 - y0).abs().max()) / scale,
                "keys": sorted(g1) == sorted(g0) and (len(g0) > 0) == grad,
                "g": max([float((g1[k] - g0[k]).abs().max()) / gscale for k in g0] or [0.0])})

        # ---- the captured all-reduce: SharePipeline replays over the one-rank group -----------------------------
        D.ALWAYS_COMMUNICATE = True
        src, dst, x = _problem(1)
        src, dst, x = src.to(dev), dst.to(dev), x.to(dev)
        m = _model(1, dev, "pwl").eval()
        part = D.VertexPartition(N, 1, 0)
        plan = D.build_halo_plan(syn.hop1_csr(src, dst, N), part)
        xc = x[plan.node_ids()].contiguous()
        with torch.no_grad():
            stacked = stack_mlps(m.fs)
            lut = m._lut_global(plan.graph)
        seen = []

        def share_forward(tables, marks=None):
            def both(name):
                seen.append(name)
                if marks is not None:
                    marks(name)
            return D.halo_recompute_forward(xc, plan, stacked, lut, True, order="reference", out_channels=1, tables=tables,
                                            marks=both)
        with torch.no_grad():
            want = share_forward(functional.TablePrefetch(stacked).launch()).clone()
        share = D.SharePipeline(share_forward, stacked, x=xc)
        got = [share.step().clone() for _ in range(6)]
        torch.cuda.synchronize()
        report["share"] = {"worst": max(float((o - want).abs().max()) for o in got) / float(want.abs().max()),
                           "tripped": share.tripped(), "collective_stage_seen": "total" in seen,
                           "kernel_nodes": [int(getattr(g, "kernel_nodes", -1)) for g in share.graphs]}
        # new weights must reach the replays (every forward consumes a table build of its own): scale the last layers
        with torch.no_grad():                       # (the stacked tensors are what the captured table builds read)
            stacked.w_last.mul_(2.0)
            if stacked.b_last is not None:
                stacked.b_last.mul_(2.0)
        after = [share.step().clone() for _ in range(3)]
        torch.cuda.synchronize()
        report["share"]["doubled"] = float((after[-1] - 2.0 * want).abs().max()) / float(want.abs().max())
        D.ALWAYS_COMMUNICATE = False
    finally:
        with open(out_path, "w") as f:
            json.dump(report, f)
        dist.destroy_process_group()


def test_every_partition_and_the_captured_all_reduce_on_a_one_rank_rccl_group(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    out = tmp_path / "report.json"
    mp.spawn(_worker, args=(_free_port(), str(out)), nprocs=1, join=True)
    rep = json.loads(out.read_text())
    assert rep["backend"] == "nccl" and rep["world"] == 1
    assert rep["collectives"] and all(rep["collectives"].values()), rep["collectives"]
    assert len(rep["cases"]) == len(CASES)
    for c in rep["cases"]:
        # identity collectives: the same kernels on the same inputs.  The inference branch of the halo forward aggregates
        # against zero column sums and adds the rest term after the all-reduce landed — another summation order
        assert c["keys"], c
        assert c["y"] <= 2e-6 and c["g"] <= 2e-6, c
    sh = rep["share"]
    assert sh["collective_stage_seen"] and not sh["tripped"], sh
    assert sh["worst"] <= 1e-6 and sh["doubled"] <= 2e-6, sh


def test_bench_force_dist_goes_through_rccl():
    """``bench.py --force-dist``: the one-GPU line initialises the ``nccl`` backend, runs the share with its collectives and
    reports the group's size; the result equals the plain one-rank line."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from test_gpu_multirank import _bench_line
    forced = _bench_line(["--gpus", "1", "--force-dist", "--partition", "halo", "--traffic", "off", "--sustain-seconds", "0"])
    plain = _bench_line(["--gpus", "1", "--traffic", "off", "--sustain-seconds", "0"])
    assert forced["ranks_seen"] == 1 and forced["process_group"] == "nccl" and forced["collectives_forced"] is True
    assert forced["share_replayed_from_hipgraphs"] is True, forced["share_graph_note"]
    assert plain.get("process_group") is None
    assert abs(forced["checksum"] - plain["checksum"]) <= 1e-5 * abs(plain["checksum"])
