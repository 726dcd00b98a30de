#!/usr/bin/env python3
"""Benchmark of the GNAN aggregation hot path on MI355X — prints ONE JSON line (rank 0).

Metric (BASELINE.json): edges aggregated / second for the TensorGNAN forward, next to the achieved HBM
bandwidth of the rho-weighted CSR SpMM.  Workload: Graph500 R-MAT scale 24 trimmed
to 10M nodes / 100M edges (+ one self pair per node), K = 1 hop codes, 64 feature columns, H = 64, L = 3,
one output channel, evaluated in the reference's order (aggregate all 64 per-feature columns, then sum
over features: models.py:373-376), fp32, synthetic data, random O(1) weights.

A "step" is one full forward over the whole graph:  table build -> shape functions -> aggregation -> feature sum.
With N > 1 ranks the node range is vertex-partitioned (strong scaling: the graph is fixed) and, in the reference
order, every rank also holds the x rows of its halo and evaluates their shape functions itself, so the only
collective is an all-reduce of the 64 column sums (gnan_amd/distributed.py: halo recompute; --partition picks the
all-gather or the feature-sharded variants instead).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 3
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--edges", type=int, default=100_000_000)
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--feat", type=int, default=64)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--out", type=int, default=1)
    ap.add_argument("--order", default="reference", choices=["reference", "sum_first"])
    ap.add_argument("--fmlp-algo", default="auto", choices=["auto", "lane", "mfma", "pwl"],
                    help="shape-function strategy: auto = exact table look-up at this size; mfma = fp32 matrix cores")
    ap.add_argument("--operand", default="f32", choices=["f32", "bf16"],
                    help="storage format of the aggregated operand rows (bf16: storage only, fp32 accumulation; the "
                         "papers100M-shaped configuration of BASELINE.json; not comparable with the fp32 reference at 1e-5)")
    ap.add_argument("--partition", default="auto", choices=["auto", "vertex", "feature", "halo", "exchange"],
                    help="multi-GPU decomposition: vertex blocks + all-gather of the operand; feature columns + all-reduce of "
                         "the [N, C] partial outputs; halo = vertex blocks, the halo's shape functions recomputed (nothing of "
                         "size N on the wire); exchange = vertex blocks, only the listed remote operand rows travel "
                         "(x stays sharded); auto = fewer bytes over xGMI")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL on ROCm); gloo + --same-device is a dry run of the "
                         "multi-rank code path on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true", help="development aid: every rank uses cuda:0")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="development aid on a 1-GPU box: run rank 0's share of a P-rank job WITHOUT the collectives "
                         "(stage times only; the printed value is not a result)")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "on", "off"],
                    help="build the shape-function tables of forward k+1 on a side stream while forward k's look-up and "
                         "aggregation run (one build per forward either way).  auto: on for a rank's share of a multi-rank job "
                         "(the 0.057-ms build is 6 %% of a 1/8 share: 1.00 -> 0.94 ms), off on one GPU (+-0 there, and the timed "
                         "call is the drop-in module's forward)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nodes", type=int, default=500_000, help="upper bound of the CPU shape-function sample")
    ap.add_argument("--cpu-rows", type=int, default=2_000_000, help="rows of the CPU aggregation sample")
    return ap.parse_args()


def spmm_algorithmic_bytes(g, W, W_out, elem=4):
    """nnz*(4 col + 1 code + W*4 gathered row) + rows*(rowptr + W_out*4 output + 12 count table);
    W_out = C when the feature sum is fused into the epilogue (reference order), else W."""
    rp = 8 if g.rowptr.dtype == torch.int64 else 4
    return g.nnz * (4 + 1 + W * elem) + g.n_rows * (rp + W_out * 4 + 4 * g.n_codes)


def git_sha():
    """Commit of the sources: from git where the checkout has a .git, else the stamp build.py left next to the library
    (the GPU boxes receive a snapshot without .git)."""
    try:
        import subprocess
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                             timeout=5).stdout.strip()
        if sha:
            return sha
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "graph-neural-additive-networks---gnan_amd", "libgnan_hip.sha")) as f:
            return f.read().strip() or None
    except OSError:
        return None


def fmlp_flops(n, F, H, L, C):
    return 2.0 * n * F * (H + max(L - 2, 0) * H * H + H * C) if L >= 2 else 2.0 * n * F * C


def fmlp_stage(args, x, H, L, C, W_out, ms):
    """The shape-function stage against the roofline that bounds the strategy in use: the exact table look-up (what AUTO
    picks at this size) streams x in and the operand rows out — HBM-bound; the matrix-core kernel is fp32-MFMA-bound."""
    if ms <= 0:
        return None
    n, F = int(x.shape[0]), int(x.shape[1])
    out_bytes = n * W_out * (2 if args.operand == "bf16" else 4)
    if args.fmlp_algo in ("auto", "pwl"):
        b = n * F * 4 + out_bytes
        return {"algo": "table look-up (pwl_build + fpwl)", "bound": "hbm", "algorithmic_bytes": b,
                "achieved_GBps": b / (ms / 1e3) / 1e9, "frac": b / (ms / 1e3) / 1e9 / HBM_PEAK_GBPS}
    fl = fmlp_flops(n, F, H, L, C)
    return {"algo": args.fmlp_algo, "bound": "mfma", "flops": fl, "achieved_TFLOPs": fl / (ms / 1e3) / 1e12,
            "frac": fl / (ms / 1e3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}


def cpu_baseline(args, model, g, x, operand_full, out_gpu):
    """Time the oracle (PyTorch-CPU restatement of the reference path) on a bounded sample of the same workload.

    Shape functions: the reference's own per-feature Python loop of nn.Linear calls (GNAN.py:58-62); the thread
    count is calibrated (all cores is not the fastest for F small GEMMs) and reported.  Aggregation:
    torch.sparse_csr (MKL) with the same weight table, all cores.  Each leg is sized to ~10 s and scaled to
    the full graph."""
    from oracle import gnan_oracle as O
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ncpu = os.cpu_count()
    x_host = x[: min(args.cpu_nodes, x.shape[0])].cpu()
    rates = {}
    for th in sorted({ncpu, min(ncpu, 64), min(ncpu, 16)}):
        torch.set_num_threads(th)
        O.feature_mlps(x_host[:256], sd)                           # warm-up (MKL thread pool, allocator)
        t0 = time.perf_counter()
        O.feature_mlps(x_host[:1500], sd)
        rates[th] = 1500 / (time.perf_counter() - t0)
    th_f = max(rates, key=rates.get)
    torch.set_num_threads(th_f)
    n_f = int(min(x_host.shape[0], max(10000, rates[th_f] * 10)))
    t0 = time.perf_counter()
    fx_cpu = O.feature_mlps(x_host[:n_f], sd)                      # Python loop over features, GNAN.py:58-62
    t_f = time.perf_counter() - t0
    # every bench run is a parity run: the oracle's sample against the rows the GPU produced for the same nodes
    fx_cpu = fx_cpu.sum(1) if args.order == "sum_first" else fx_cpu.reshape(n_f, -1)
    fx_gpu = operand_full[:n_f].float().cpu()[:, : fx_cpu.shape[1]]
    parity_f = float((fx_gpu.double() - fx_cpu.double()).abs().max() / fx_cpu.double().abs().max())

    torch.set_num_threads(ncpu)
    n_r = min(args.cpu_rows, g.n_rows)
    rowptr = g.rowptr[: n_r + 1].cpu().long().numpy()
    nnz_s = int(rowptr[-1])
    col = g.col[:nnz_s].cpu().numpy()
    code = g.code[:nnz_s].cpu().numpy()
    cnt = g.cnt[:n_r].cpu().long().numpy()
    S = operand_full.cpu()
    lut = O.rho_lut(sd, g.n_codes)
    O.spmm_csr_sparse(rowptr[:1001], col[: rowptr[1000]], code[: rowptr[1000]], S, lut, cnt[:1000])   # warm-up
    t0 = time.perf_counter()
    y = O.spmm_csr_sparse(rowptr, col, code, S, lut, cnt)
    y = y.sum(dim=1)
    t_s = time.perf_counter() - t0
    # parity of the aggregation: the GPU's output rows against the same restatement in FLOAT64 on a slice of the sample
    # (the float32 host run above is itself ~2e-4 off: torch's CPU column sum of 10^7 float32 rows for the rest bucket)
    n_p = min(n_r, 250_000)
    nnz_p = int(rowptr[n_p])
    y64 = O.spmm_csr_sparse(rowptr[: n_p + 1], col[:nnz_p], code[:nnz_p], S.double(), lut.double(), cnt[:n_p]).sum(dim=1)
    got = out_gpu[:n_p].float().cpu().reshape(n_p, -1).sum(1).double()
    parity_s = float((got - y64).abs().max() / y64.abs().max())
    parity_cpu32 = float((y[:n_p].double() - y64).abs().max() / y64.abs().max())
    n_tot, nnz_tot = args.nodes, args.edges + args.nodes
    est = t_f * n_tot / n_f + t_s * nnz_tot / nnz_s
    return {
        # cores: the most threads any leg used (the shape-function loop runs fastest on fewer: fmlp_cores)
        "value": args.edges / est, "unit": "edges/s", "cores": max(th_f, ncpu), "kind": "port",
        "sample": (f"oracle/gnan_oracle.py on the host ({ncpu} hardware threads): shape functions on the first {n_f} "
                   f"nodes with {th_f} threads ({t_f:.2f} s; calibrated nodes/s by threads: "
                   f"{ {k: round(v) for k, v in rates.items()} }) + torch.sparse_csr aggregation of the first {n_r} "
                   f"rows / {nnz_s} pairs against the full {S.shape[0]}x{S.shape[1]} operand with {ncpu} threads "
                   f"({t_s:.2f} s); both legs scaled to the full graph"),
        "fmlp_nodes_per_s": n_f / t_f, "spmm_pairs_per_s": nnz_s / t_s, "fmlp_cores": th_f, "spmm_cores": ncpu,
        # max |gpu - oracle| / max |oracle| on sampled rows: shape functions of the first n_f nodes (float32 oracle),
        # aggregated output of the first n_p rows (torch.sparse_csr in float64 over the GPU's operand rows)
        "parity_max_rel_err": max(parity_f, parity_s), "parity_fmlp_rel_err": parity_f, "parity_spmm_rel_err": parity_s,
        "parity_rows": {"fmlp": n_f, "spmm": n_p}, "cpu_f32_vs_f64_rel_err": parity_cpu32,
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import gnan_amd  # noqa: F401
    from gnan_amd import synthetic as syn
    from gnan_amd.distributed import (FeaturePartition, VertexPartition, build_exchange_plan, build_halo_plan,
                                      choose_partition, feature_parallel_forward, halo_exchange_forward,
                                      halo_recompute_forward, partitioned_forward, slice_features)
    from gnan_amd import _lib, functional
    from gnan_amd.functional import stack_mlps
    from gnan_amd.graph import hop_inputs
    functional.FMLP_ALGO = {"auto": _lib.FMLP_AUTO, "lane": _lib.FMLP_LANE, "mfma": _lib.FMLP_MFMA,
                            "pwl": _lib.FMLP_PWL}[args.fmlp_algo]
    from gnan_amd.models import TensorGNAN

    N, E, F, H, L, C = args.nodes, args.edges, args.feat, args.hidden, args.layers, args.out
    emulated = args.emulate_world > 1 and world == 1
    pworld = args.emulate_world if emulated else world       # how many shares the work is cut into
    args.pipeline = args.pipeline == "on" or (args.pipeline == "auto" and pworld > 1)
    partition = args.partition if args.partition != "auto" else choose_partition(N, F, C, pworld, args.order)
    if emulated and partition in ("vertex", "exchange"):
        raise SystemExit("--emulate-world covers the halo and feature partitions (these shares need the other ranks' operand rows)")
    if partition == "feature" and args.order != "reference":
        raise SystemExit("--partition feature needs --order reference (sum-first exchanges the narrow operand)")
    part = VertexPartition(N, pworld, rank)
    fpart = FeaturePartition(F, pworld, rank)
    if emulated:                                             # same shares, no process group: collectives are skipped
        part.world = fpart.world = 1
        part.__class__ = type("EmuV", (VertexPartition,), {"block": property(lambda self: -(-N // pworld))})
        fpart.__class__ = type("EmuF", (FeaturePartition,), {"block": property(lambda self: -(-F // pworld))})
    t_setup = time.perf_counter()
    src, dst = syn.rmat_edges(args.scale, N, E, seed=0, device=dev)
    plan = None
    if partition == "halo":                      # owned rows; x rows of the owned nodes AND of the remote nodes they list
        plan = build_halo_plan(syn.hop1_csr(src, dst, N, part.lo, part.hi), part)
        g = plan.graph
        x = syn.block_features(N, F, 0, N, seed=1, device=dev)[plan.node_ids()].contiguous()
    elif partition == "exchange":                # owned rows only; the listed remote operand rows arrive per forward
        xplan = build_exchange_plan(syn.hop1_csr(src, dst, N, part.lo, part.hi), part)
        g = xplan.halo.graph
        x = syn.block_features(N, F, part.lo, part.hi, seed=1, device=dev)
    elif partition == "vertex":
        g = syn.hop1_csr(src, dst, N, part.lo, part.hi)
        x = syn.block_features(N, F, part.lo, part.hi, seed=1, device=dev)
    else:                                        # whole graph, this rank's feature columns
        g = syn.hop1_csr(src, dst, N)
        x = syn.block_features(N, F, 0, N, seed=1, device=dev)[:, fpart.lo:fpart.hi].contiguous()
    del src, dst
    g.long_row_plan()
    torch.manual_seed(0)
    model = TensorGNAN(F, C, L, hidden_channels=H, normalize_rho=True, rho_per_feature=False, device="cuda")
    with torch.no_grad():                                    # O(1)-scale weights (the upstream init gives ~1e-14 outputs)
        for p in model.parameters():
            if p.dim() == 2:
                torch.nn.init.xavier_normal_(p, gain=1.0)
            else:
                p.normal_(0.0, 0.5)
    model = model.to(dev).eval()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    with torch.no_grad():
        stacked = stack_mlps(model.fs)
        lut = model.rho(hop_inputs(g.n_codes, dev).view(-1, 1))
    op_dtype = torch.bfloat16 if args.operand == "bf16" else torch.float32
    stage_names = ["fmlp", "gather", "total", "spmm"] + (["reduce"] if partition == "feature" else [])
    # One GPU: the timed call is the drop-in itself, TensorGNAN.forward(data) (models.py:358-384) on a data object that
    # carries the hop-coded graph — graph lookup, weight views, rho on the D distinct distances, look-up, aggregation.
    # More than one rank: gnan_amd.distributed's forward of this rank's share (the reference has no multi-process path).
    use_module = world == 1 and not emulated and partition == "vertex" and not args.pipeline
    if use_module:
        class Bag:
            pass
        data = Bag()
        data.x, data.edge_index, data.gnan_graph = x, None, g
        model.aggregation_order = args.order
        model.operand_dtype = op_dtype
        stage_names = ["lut", "fmlp", "spmm"]
    stacked_local = slice_features(stacked, fpart.lo, fpart.hi) if partition == "feature" else None
    events = []
    # Software pipeline of the inference loop: the table build (64 workgroups, 0.05 ms, a function of the weights only) of
    # the NEXT forward runs on a side stream under the current forward's look-up and aggregation.  Every forward consumes
    # a build of its own; nothing is cached.  (halo and vertex partitions; the feature partition builds in line.)
    prefetch = None
    if args.pipeline and partition in ("halo", "vertex") and args.fmlp_algo in ("auto", "pwl"):
        prefetch = functional.TablePrefetch(stacked)
        if not prefetch.applies:
            prefetch = None
    pipe = {"next": prefetch.launch() if prefetch else None}

    def step(record):
        marks = {}
        if record:
            def mark(name):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks[name] = ev
        else:
            mark = None
        tables = None
        if prefetch is not None:
            tables, pipe["next"] = pipe["next"], prefetch.launch()      # this forward's tables; the next forward's build starts now
        with torch.no_grad():
            if use_module:
                model.stage_hook = mark
                out = model.forward(data)
                model.stage_hook = None
            elif partition == "halo":
                out = halo_recompute_forward(x, plan, stacked, lut, True, order=args.order, out_channels=C,
                                             marks=mark, operand_dtype=op_dtype, tables=tables)
            elif partition == "exchange":
                out = halo_exchange_forward(x, xplan, stacked, lut, True, order=args.order, out_channels=C,
                                            marks=mark, operand_dtype=op_dtype)
            elif partition == "vertex":
                out = partitioned_forward(x, g, stacked, lut, True, part, order=args.order, out_channels=C,
                                          marks=mark, operand_dtype=op_dtype, tables=tables)
            else:
                out = feature_parallel_forward(x, g, stacked_local, lut, True, fpart, out_channels=C, marks=mark,
                                               operand_dtype=op_dtype)
        if record:
            events.append(marks)
        return out

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step(False)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    checksum = out.double().sum().reshape(1)
    if world > 1 and partition != "feature":      # row-partitioned output: add the ranks' shares (feature: already whole)
        dist.all_reduce(checksum, op=dist.ReduceOp.SUM)
    checksum = float(checksum)

    stages = {n: 0.0 for n in stage_names}
    per_step = []                                    # device time of every timed step (HIP events on the compute stream)
    for m in events:
        prev = m["start"]
        for n in stage_names:                        # stages that are empty by construction record no event (0 ms)
            if n in m:
                stages[n] += prev.elapsed_time(m[n])
                prev = m[n]
        per_step.append(m["start"].elapsed_time(m[stage_names[-1]]))
    stages = {n: v / max(1, len(events)) for n, v in stages.items()}
    per_step.sort()

    if partition == "feature":
        W = (fpart.hi - fpart.lo) * C
    else:
        W = F * C if args.order == "reference" else C
    b_alg = spmm_algorithmic_bytes(g, W, C, 2 if args.operand == "bf16" else 4)
    spmm_s = stages["spmm"] / 1e3
    achieved = b_alg / spmm_s / 1e9 if spmm_s > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    workload = f"rmat_s{args.scale}_{N}n_{E}e_F{F}_H{H}_L{L}_C{C}_{args.order}_K1" + ("" if args.operand == "f32" else "_bf16")
    traffic_source = None
    if os.path.exists(tpath):
        rec = json.load(open(tpath))
        if rec.get("workload") == workload and rec.get("n_gpus") == world:
            traffic = rec.get("bytes_per_launch")
            # PMC counters need rocprofv3 around the process: the figure is the committed one of the same command, not of this run
            traffic_source = "profiles/hbm_traffic.json (%s; rocprofv3 --pmc passes of this command, not measured in this run)" % rec.get("source", "committed")

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        result = {
            "metric": "edges aggregated/sec, TensorGNAN forward", "value": E / (elapsed / args.steps),
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.operand == "f32" else "bf16 operand storage, f32 accumulate",
            "data": "synthetic",
            "config": {"workload": workload, "nodes": N, "edges": E, "stored_pairs_rank0": g.nnz,
                       "operand_width": W, "partition": f"{partition} x{world}", "exchange":
                       "none" if world == 1 else ("all_reduce(column sums [W])" if partition == "halo" else
                                                  "all_gather(operand [N,W])" if partition == "vertex" else
                                                  "all_to_all_v(listed remote operand rows [n_halo,W]) + all_reduce(column sums [W])"
                                                  if partition == "exchange" else "all_reduce(out [N,C])")},
            "roofline": {"bound": "hbm", "kernel": "spmm_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": b_alg, "avg_launch_ms": stages["spmm"]},
            "stages_ms": stages,
            "step_ms_device": {"min": per_step[0], "median": per_step[len(per_step) // 2]} if per_step else None,
            "seeds": {"graph": 0, "features": 1, "weights": 0}, "git_sha": git_sha(),
            "spmm_edges_per_s": (g.nnz - g.n_rows) / spmm_s if spmm_s > 0 else None,
            "fmlp": fmlp_stage(args, x, H, L, C, W if args.order == "reference" else C, stages["fmlp"]),
            "pipelined_table_build": prefetch is not None,
            "timed_call": "gnan_amd.models.TensorGNAN.forward(data)" if use_module else f"gnan_amd.distributed ({partition})",
            "emulated_share_of": pworld if emulated else None, "setup_s": t_setup, "checksum": checksum,
            "operand_rows_rank0": int(x.shape[0]),
        }
        if world == 1 and not args.no_cpu_baseline:
            with torch.no_grad():
                from gnan_amd.functional import feature_mlps
                operand = feature_mlps(x, stacked, args.order == "sum_first")
            result["cpu_baseline"] = cpu_baseline(args, model, g, x, operand, out)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
