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
    ap.add_argument("--config", default="c4", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configuration (SURVEY.md section 8d): c4 = R-MAT 10M nodes / 100M edges, the headline "
                         "(default); c5 = papers100M-shaped, bf16 operand rows; c3 = ogbn-arxiv-shaped forward+backward; "
                         "c2 = Mutagenicity-shaped graph-level task (one small graph per forward)")
    ap.add_argument("--graphs", type=int, default=4337, help="c2: number of graphs (Mutagenicity has 4337)")
    ap.add_argument("--loop", default="harness", choices=["harness", "reference"],
                    help="c2 / c3: 'reference' adds the training step as the reference's UNCHANGED loop issues it (tests/reference_loop.py "
                         "restates trainer.py:23-86: anomaly mode, zero_grad, forward, mask, loss, backward, stock Adam over "
                         "model.parameters(), loss.item()): eager ms per step, with and without anomaly mode")
    ap.add_argument("--fresh-inputs", action="store_true",
                    help="--loop reference: a second leg whose batches live on the host — data.to(device) builds NEW device tensors "
                         "every step (trainer.py:46), the upload timed on its own as well")
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
    ap.add_argument("--cut", default="cost", choices=["cost", "rows"],
                    help="halo / exchange partitions: node blocks of equal cost (look-up rows + stored pairs, "
                         "distributed.balanced_bounds) or of equal row count")
    ap.add_argument("--alt-timeout", type=float, default=300.0,
                    help="seconds the extra partitions may take before the measured line is printed without them")
    ap.add_argument("--alt-partitions", default="auto", choices=["auto", "off"],
                    help="more than one rank, halo partition: after the timed loop, also time a few steps of the partitions that DO "
                         "move operand rows over xGMI (all-gather of the [N, W] operand; all-to-all-v of the listed halo rows) and "
                         "print them as alt_partitions")
    ap.add_argument("--alt-steps", type=int, default=5)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL on ROCm); gloo + --same-device is a dry run of the "
                         "multi-rank code path on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true", help="development aid: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="one rank: still create the process group on --backend and run the share's collectives over it "
                         "(distributed.ALWAYS_COMMUNICATE) — RCCL on a 1-GPU box; combines with --emulate-world")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="development aid on a 1-GPU box: run rank 0's share of a P-rank job WITHOUT the collectives "
                         "(stage times only; the printed value is not a result)")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "on", "off"],
                    help="build the shape-function tables of forward k+1 on a side stream while forward k's look-up and "
                         "aggregation run (one build per forward either way).  auto: on for a rank's share of a multi-rank job "
                         "(the 0.057-ms build is 6 %% of a 1/8 share: 1.00 -> 0.94 ms), off on one GPU (+-0 there, and the timed "
                         "call is the drop-in module's forward)")
    ap.add_argument("--share-graph", default="auto", choices=["auto", "on", "off"],
                    help="replay a rank's share of the forward from two alternating hipGraphs (distributed.SharePipeline: one "
                         "graph launch per forward, the next forward's table build on a forked branch, the all-reduce of the "
                         "column sums captured with RCCL).  auto: on where --pipeline is on (a multi-rank share), falling back "
                         "to the eager loop if the step cannot be captured or its first replay disagrees with the eager forward")
    ap.add_argument("--share-fork", default="fmlp", choices=["start", "fmlp"],
                    help="where the next forward's table build branches off inside a replayed share: under the look-up or "
                         "under the aggregation (default: measured 0.919 against 0.944 ms on the slowest 1/8 share)")
    ap.add_argument("--index-buckets", type=int, default=0, help="cells per feature of the direct-index look-up (0: the library's default)")
    ap.add_argument("--sustain-seconds", type=float, default=10.0,
                    help="one GPU: after the timed region (and the CPU baseline) keep issuing the same step for this long, so "
                         "that an outside sampler of GPU activity sees the kernels (the timed region is ~0.1 s of a run whose "
                         "wall time is mostly the CPU baseline); reported as sustained_ms_per_step, never as `value`.  0: off")
    ap.add_argument("--traffic", default="measure", choices=["measure", "committed", "off"],
                    help="roofline.traffic: measure = two short rocprofv3 --pmc child runs of this very command (FETCH_SIZE, "
                         "WRITE_SIZE; one GPU), falling back to the committed figure of profiles/hbm_traffic.json; committed = "
                         "that figure only")
    ap.add_argument("--set", action="append", default=[], metavar="MODULE.NAME=VALUE",
                    help="A/B aid: set a module-level constant of the package before the run, e.g. --set "
                         "aggregate.HOT_ROWS_IN_LDS=False (the library and the package read no environment switches for "
                         "kernel selection)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nodes", type=int, default=500_000, help="upper bound of the CPU shape-function sample")
    ap.add_argument("--cpu-rows", type=int, default=2_000_000, help="rows of the CPU aggregation sample")
    args = ap.parse_args()
    if args.config == "c5":                                  # papers100M-shaped: same generator, scale 27, bf16 operand rows
        given = set(a.split("=")[0] for a in sys.argv[1:])
        if "--scale" not in given:
            args.scale = 27
        if "--nodes" not in given:
            args.nodes = 111_059_956
        if "--edges" not in given:
            args.edges = 1_615_685_872
        if "--operand" not in given:
            args.operand = "bf16"
    if args.config == "c3":
        given = set(a.split("=")[0] for a in sys.argv[1:])
        if "--nodes" not in given:
            args.nodes = 169_343
        if "--edges" not in given:
            args.edges = 1_166_243
        if "--feat" not in given:
            args.feat = 129
    return args


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


def cpu_model():
    """Model string of the host CPU (SURVEY.md section 8d asks for it next to the core count)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
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
        # cores: the threads of the DOMINANT leg of the estimate (the shape-function loop, which runs fastest on few threads:
        # fmlp_cores; the aggregation leg ran on spmm_cores); dominant_leg_share says how dominant
        "value": args.edges / est, "unit": "edges/s",
        "cores": th_f if t_f * n_tot / n_f >= t_s * nnz_tot / nnz_s else ncpu, "kind": "port", "cpu_model": cpu_model(),
        "dominant_leg": "shape functions" if t_f * n_tot / n_f >= t_s * nnz_tot / nnz_s else "aggregation",
        "dominant_leg_share": max(t_f * n_tot / n_f, t_s * nnz_tot / nnz_s) / est, "hardware_threads": ncpu,
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


def _redraw(model):
    """O(1)-scale weights (the upstream init, xavier with gain 0.01, gives ~1e-14 outputs)."""
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.dim() == 2:
                torch.nn.init.xavier_normal_(p, gain=1.0)
            else:
                p.normal_(0.0, 0.5)


class _Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self


class _HostBag:
    """A batch as the reference's loader holds it — on the HOST (pinned): ``to(device)`` builds NEW device tensors every call, what
    trainer.py:46 does per step (datasets.py:339-349 collates, ``data.to(device)`` uploads).  Attributes that are not tensors (an
    attached ``gnan_graph``) travel as they are."""

    def __init__(self, **kw):
        self.__dict__.update({k: (v.detach().cpu().pin_memory() if torch.is_tensor(v) else v) for k, v in kw.items()})

    def to(self, device):
        return _Bag(**{k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})


def _upload_ms(batches, device, reps=3):
    """ms per batch of ``data.to(device)`` alone (the H2D copy the reference's loop pays every step)."""
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            b.to(device)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / len(batches) * 1e3
        best = t if best is None else min(best, t)
    return best


def _event_ms(fn, steps, warmup):
    """Device time of ``fn`` per call: HIP events on torch's current stream (the stream every kernel of the library is
    launched on), median and min over ``steps`` calls after ``warmup``."""
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def apply_sets(args):
    """--set MODULE.NAME=VALUE: module-level constants of the package, before the run (A/B aid)."""
    import ast
    import importlib
    for item in args.set:
        target, value = item.split("=", 1)
        mod_name, attr = target.rsplit(".", 1)
        mod = importlib.import_module("gnan_amd." + mod_name)
        if not hasattr(mod, attr):
            raise SystemExit(f"--set: gnan_amd.{mod_name} has no {attr}")
        setattr(mod, attr, ast.literal_eval(value))


def reference_loop_leg(model, batches, n_out, graph_task, epochs=4, fresh_inputs=False):
    """ms per step of the reference-shaped training loop over ``batches`` on a deep copy of ``model`` (stock Adam, eager).
    ``fresh_inputs``: the batches live on the host and ``data.to(device)`` builds new device tensors every step, exactly as
    trainer.py:46 after datasets.py:339-349 — the figure then includes that upload, which is reported on its own as well."""
    import copy
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import reference_loop
    loss_fn = torch.nn.BCEWithLogitsLoss() if n_out == 1 else torch.nn.CrossEntropyLoss()
    out = {}
    device = batches[0].x.device
    if fresh_inputs:
        # (the loop's label handling, trainer.py:32-40, now runs on host tensors: on a 256-thread host a 169k-element comparison costs
        #  tens of ms of OpenMP start-up — the host side is limited as tests/conftest.py limits it)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        hosts = {}
        for b in batches:                               # (a repeated batch: one host copy)
            if id(b) not in hosts:
                hosts[id(b)] = _HostBag(**b.__dict__)
        batches = [hosts[id(b)] for b in batches]
        out["upload_ms_per_step"] = _upload_ms(batches, device)
        epochs = max(epochs, 4)                         # the inputs are adopted on the third step and captured three steps later
    import gnan_amd
    # the optimizer as main.py:141 builds it — torch.optim.Adam(model.parameters()): the F x L per-layer tensors, torch's own
    # per-tensor bookkeeping included — with and without anomaly mode; and over gnan_amd.optim_params(model), the flat buffers those
    # tensors are views of (a one-line change in main.py; same numbers, a dozen tensors)
    for tag, anomaly, flat in (("anomaly_mode", True, False), ("plain", False, False), ("optim_params_plain", False, True)):
        twin = copy.deepcopy(model).eval()
        opt = torch.optim.Adam(gnan_amd.optim_params(twin) if flat else twin.parameters(), lr=1e-3)
        best = None
        for _ in range(epochs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = reference_loop.train_epoch(twin, batches, loss_fn, opt, device, classify=True,
                                             is_graph_task=graph_task, detect_anomaly=anomaly)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / len(batches) * 1e3
            best = t if best is None else min(best, t)
        out[f"{tag}_ms_per_step"] = best
        if fresh_inputs:
            out[f"{tag}_ms_per_step_without_upload"] = best - out["upload_ms_per_step"]
        out[f"{tag}_last_loss"] = float(ret[0])
        out["optimizer_tensors_flat" if flat else "optimizer_tensors"] = sum(len(g["params"]) for g in opt.param_groups)
    out["loop"] = ("tests/reference_loop.train_epoch (trainer.py:23-86 restated: set_detect_anomaly, zero_grad, forward, mask, loss, "
                   "backward, torch.optim.Adam(model.parameters()).step(), loss.item() per step), best of %d epochs" % epochs)
    return out


def run_c3(args):
    """BASELINE config 3: ogbn-arxiv-shaped TensorGNAN forward + backward on one GPU (SURVEY.md section 8d C3;
    /root/reference datasets.py:273-291: N = 169 343, E = 1 166 243, 128 features + the ones column, num_classes = 1 as the
    reference sets it; --out 40 = the data set's true class count).  Preferential-attachment edges (seed 0) + one self pair
    per node, K = 1 hop codes, rest bucket on; models.TensorGNAN in its default (sum-first) order, fp32.
    A step = forward + loss + backward of the drop-in module (eager autograd, gradients reset as trainer.py:66 does)."""
    import gnan_amd  # noqa: F401
    from gnan_amd import synthetic as syn
    from gnan_amd.models import TensorGNAN
    apply_sets(args)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    N, E, F, H, L, C = args.nodes, args.edges, args.feat, args.hidden, args.layers, args.out
    t_setup = time.perf_counter()
    src, dst = syn.preferential_attachment_edges(N, E, seed=0, device=dev)
    g = syn.hop1_csr(src, dst, N)
    x = syn.block_features(N, F, 0, N, seed=1, device=dev)
    torch.manual_seed(0)
    model = TensorGNAN(F, C, L, hidden_channels=H, normalize_rho=True, rho_per_feature=False, device="cuda")
    _redraw(model)
    model = model.to(dev).eval()
    data = _Bag(x=x, edge_index=None, gnan_graph=g)
    gen = torch.Generator(device=dev).manual_seed(2)
    target = torch.randn(N, C, generator=gen, device=dev)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    marks_all = []

    def fwd(record=False):
        marks = {}
        if record:
            def mark(name):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks[name] = ev
            model.stage_hook = mark
        with torch.no_grad():
            out = model.forward(data)
        model.stage_hook = None
        if record:
            marks_all.append(marks)
        return out

    loss_fn = torch.nn.MSELoss()
    params = list(model.parameters())            # (as an optimizer holds them: walking the module tree costs 0.7 ms per call)

    def fwd_bwd():
        for p in params:                         # what optimizer.zero_grad() of an optimizer over model.parameters() does (trainer.py:48)
            p.grad = None
        out = model.forward(data)
        loss = loss_fn(out, target)              # a torch loss module, as the trainer's loss_fn(outputs, labels) (trainer.py:61-64)
        loss.backward()
        return out

    for _ in range(args.warmup):
        fwd()
        fwd_bwd()
    fwd_ms, fwd_min = _event_ms(lambda: fwd(True), args.steps, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = fwd_bwd()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fb_ms, fb_min = _event_ms(fwd_bwd, args.steps, 0)
    # ... and the same step replayed from ONE hipGraph launch (what gnan_amd.harness does with a full-batch epoch from its
    # third pass on): the eager step above is bound by the host issuing ~60 launches, not by the device
    replay = {"replayed_fwd_bwd_ms": None, "replayed_fwd_bwd_ms_min": None, "replay_note": None}
    try:
        from gnan_amd.graphed import CaptureFailed, GraphedCallable
        eager_grads = [p.grad.detach().clone() for p in params]

        def clear():
            model.zero_grad()

        def static_step():
            out = model.forward(data)
            loss_fn(out, target).backward()
            return out
        captured = GraphedCallable(static_step, warmup=2, before_capture=clear)
        captured.replay()
        captured.replay()
        torch.cuda.synchronize()
        scale = max(float(g0.abs().max()) for g0 in eager_grads)
        worst = max(float((p.grad - g0).abs().max()) for p, g0 in zip(params, eager_grads))
        if not worst <= 1e-5 * scale:
            replay["replay_note"] = f"replayed gradients off by {worst / max(scale, 1e-30):.2e} of the largest: not reported"
        else:
            replay["replayed_fwd_bwd_ms"], replay["replayed_fwd_bwd_ms_min"] = _event_ms(captured.replay, args.steps, 0)
            replay["replay_note"] = (f"{captured.kernel_nodes} kernels per replay; gradients within {worst / max(scale, 1e-30):.1e} "
                                     "of the eager step's")
    except (CaptureFailed, RuntimeError) as e:
        replay["replay_note"] = f"not capturable ({type(e).__name__}: {str(e)[:160]})"

    stage_names = ["lut", "fmlp", "spmm"]
    stages = {n: 0.0 for n in stage_names}
    for m in marks_all:
        prev = m["start"]
        for n in stage_names:
            if n in m:
                stages[n] += prev.elapsed_time(m[n])
                prev = m[n]
    stages = {n: v / max(1, len(marks_all)) for n, v in stages.items()}
    # dominant kernel of the forward: the shape-function look-up with the feature sum (x in, [N, C] out) — HBM-bound by
    # bytes, latency-bound at this size (87 MB of x): SURVEY section 8d expects no meaningful fraction here
    b_fmlp = N * F * 4 + N * C * 4
    b_spmm = spmm_algorithmic_bytes(g, C, C)
    result = {
        "metric": "edges aggregated/sec, TensorGNAN forward+backward", "value": E / (elapsed / args.steps), "unit": "edges/s",
        "n_gpus": 1, "ranks_seen": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"c3_arxiv_shaped_pa_{N}n_{E}e_F{F}_H{H}_L{L}_C{C}_sum_first_K1", "nodes": N, "edges": E,
                   "stored_pairs": g.nnz, "step": "forward + MSE loss + backward of models.TensorGNAN as a caller's loop issues them "
                                                  "(from its third step on the module replays its forward and backward from two hipGraphs: "
                                                  "gnan_amd.replay)"},
        "fwd_ms": fwd_ms, "fwd_ms_min": fwd_min, "fwd_bwd_ms": fb_ms, "fwd_bwd_ms_min": fb_min,
        "fwd_edges_per_s": E / (fwd_ms / 1e3),
        "roofline": {"bound": "hbm", "kernel": "fpwl_index_kernel (shape-function look-up, feature sum; + sum_groups_kernel)",
                     "achieved": b_fmlp / (stages["fmlp"] / 1e3) / 1e9 if stages["fmlp"] > 0 else 0.0, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": b_fmlp / (stages["fmlp"] / 1e3) / 1e9 / HBM_PEAK_GBPS if stages["fmlp"] > 0 else 0.0,
                     "traffic": None, "algorithmic_bytes_per_launch": b_fmlp, "avg_launch_ms": stages["fmlp"],
                     "note": "stage time incl. the table build; 87 MB of x: latency-bound, not bandwidth-bound"},
        "spmm_roofline": {"bound": "hbm", "kernel": "spmm_kernel<1,...> (narrow rows)", "algorithmic_bytes_per_launch": b_spmm,
                          "avg_launch_ms": stages["spmm"],
                          "achieved": b_spmm / (stages["spmm"] / 1e3) / 1e9 if stages["spmm"] > 0 else 0.0,
                          "frac": b_spmm / (stages["spmm"] / 1e3) / 1e9 / HBM_PEAK_GBPS if stages["spmm"] > 0 else 0.0},
        "stages_ms": stages, "seeds": {"graph": 0, "features": 1, "weights": 0, "target": 2}, "git_sha": git_sha(),
        "timed_call": "gnan_amd.models.TensorGNAN.forward(data) + loss.backward()", "setup_s": t_setup,
        "checksum": float(out.detach().double().sum()), "backward": True,
    }
    result.update(replay)
    if args.loop == "reference":
        gen_c = torch.Generator().manual_seed(3)
        labelled = _Bag(x=x, edge_index=None, gnan_graph=g, y=torch.randint(0, max(C, 2), (N,), generator=gen_c).to(dev),
                        train_mask=(torch.rand(N, generator=gen_c) < 0.6).to(dev))
        result["reference_loop"] = reference_loop_leg(model, [labelled] * 20, C, False, epochs=3)
        if args.fresh_inputs:
            result["reference_loop_fresh_inputs"] = reference_loop_leg(model, [labelled] * 20, C, False, epochs=3, fresh_inputs=True)
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_c3(args, model, g, x, fwd())
    print(json.dumps(result), flush=True)


def cpu_baseline_c3(args, model, g, x, out_gpu):
    """The oracle's forward on a bounded sample: the per-feature Python loop of nn.Linear calls (GNAN.py:58-62) on the first
    rows, torch.sparse_csr aggregation of all rows; and the parity of the GPU's output on sampled rows (float64)."""
    from oracle import gnan_oracle as O
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ncpu = os.cpu_count()
    torch.set_num_threads(min(ncpu, 16))
    N, F = x.shape
    n_f = min(N, 20000)
    xh = x.cpu()
    O.feature_mlps(xh[:256], sd)
    t0 = time.perf_counter()
    O.feature_mlps(xh[:n_f], sd)
    t_f = time.perf_counter() - t0
    # parity: the whole forward in float64 on the rows of a node sample (their neighbours' shape functions included)
    p64 = {k: v.double() for k, v in sd.items()}
    rowptr = g.rowptr.cpu().long().numpy()
    col, code, cnt = g.col.cpu().numpy(), g.code.cpu().numpy(), g.cnt.cpu().long().numpy()
    S64 = torch.cat([O.feature_mlps(xh[i:i + 8192].double(), p64).sum(1) for i in range(0, N, 8192)])      # [N, C] float64
    lut64 = O.rho_lut(p64, g.n_codes, dtype=torch.float64)
    torch.set_num_threads(ncpu)
    t0 = time.perf_counter()
    y32 = O.spmm_csr_sparse(rowptr, col, code, S64.float(), lut64.float(), cnt)
    t_s = time.perf_counter() - t0
    y64 = O.spmm_csr_sparse(rowptr, col, code, S64, lut64, cnt)
    parity = float((out_gpu.double().cpu() - y64).abs().max() / y64.abs().max())
    est = t_f * N / n_f + t_s
    return {"value": args.edges / est, "unit": "edges/s", "cores": ncpu, "kind": "port", "cpu_model": cpu_model(),
            "sample": (f"oracle/gnan_oracle.py forward only: shape functions on the first {n_f} of {N} nodes with "
                       f"{min(ncpu, 16)} threads ({t_f:.2f} s, scaled) + torch.sparse_csr aggregation of all rows with {ncpu} "
                       f"threads ({t_s:.2f} s)"),
            "parity_max_rel_err": parity, "parity_rows": N, "cpu_f32_vs_f64_rel_err":
            float((y32.double() - y64).abs().max() / y64.abs().max())}


def run_c2(args, rank=0, world=1):
    """BASELINE config 2: Mutagenicity-shaped graph-level task, one small graph per forward (trainer.py:23-86 feeds
    batch_size = 1; SURVEY.md section 8d C2: 4337 graphs, N_g ~ clip(round(LogNormal(3.3, 0.45)), 4, 417), F = 14 + 1, C = 1,
    models.TensorGNAN(is_graph_task=True)), dense inputs as pre_process_datasets.py:104-142 emits them, fp32.
    A step = one evaluation pass over all graphs (forward of every graph).  An 'edge' is one (i, j) pair of a graph's dense
    N_g x N_g aggregation.  More than one rank: replicas only — the graphs are dealt out round-robin, no collective."""
    import gnan_amd  # noqa: F401
    from gnan_amd import HopGraph
    from gnan_amd import synthetic as syn
    from gnan_amd.models import TensorGNAN
    import torch.distributed as dist
    dev = torch.device("cuda", 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    t_setup = time.perf_counter()
    graphs, pairs, host = [], 0, []
    for i, (ei, x, y) in enumerate(syn.mutagenicity_shaped_graphs(args.graphs, seed=0)):
        if i % world != rank:
            continue
        n = x.shape[0]
        hg = HopGraph.from_edge_index(torch.as_tensor(ei).to(dev), n)           # all-pairs BFS on the device (csrc/bfs.hip)
        code = hg.code.long()
        nd = torch.where(code == 255, torch.zeros((), device=dev), 1.0 / (1.0 + code.float()))
        norm = torch.gather(hg.cnt.float(), 1, code.clamp_max(hg.n_codes - 1))
        graphs.append(_Bag(x=x.to(dev), edge_index=None, node_distances=nd, normalization_matrix=norm,
                           y=torch.tensor([[y]], device=dev)))
        pairs += n * n
        if len(host) < 64:
            host.append((ei, x, n))
    torch.manual_seed(0)
    model = TensorGNAN(15, 1, args.layers, hidden_channels=args.hidden, is_graph_task=True, readout_n_layers=0, device="cuda")
    _redraw(model)
    model = model.to(dev).eval()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def epoch():
        with torch.no_grad():
            return [model.forward(d) for d in graphs]

    for _ in range(args.warmup):
        epoch()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = epoch()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    total_pairs = pairs
    if world > 1:
        t = torch.tensor([elapsed, float(pairs)], device=dev, dtype=torch.float64)
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, total_pairs = float(tm[0]), int(t[1])
    if rank != 0:
        return
    # replayed per shape through the harness (what a training run does after every epoch: trainer.py:89-154)
    from gnan_amd import harness
    loss_fn = torch.nn.BCEWithLogitsLoss()
    replay_ms = None
    try:
        for _ in range(3):
            harness.test_epoch(model, graphs, loss_fn, dev, classify=True, val_mask=True, is_graph_task=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        harness.test_epoch(model, graphs, loss_fn, dev, classify=True, val_mask=True, is_graph_task=True)
        torch.cuda.synchronize()
        replay_ms = (time.perf_counter() - t0) / len(graphs) * 1e3
    except Exception as e:                                   # the replayed pass is an extra figure, not the timed call
        replay_ms = f"failed: {type(e).__name__}: {e}"
    # ... and a training epoch (trainer.py:23-86: batch_size 1, forward + loss + backward + Adam per graph) on a twin of the
    # model: eager in its first passes, one captured step per graph shape from the third on
    train_ms, train_kernels, captured_steps = None, None, None
    try:
        import copy
        twin = copy.deepcopy(model).train()
        opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
        labelled = graphs                            # (they carry their labels: y [1, 1])
        for _ in range(3):
            harness.train_epoch(twin, labelled, loss_fn, opt, dev, classify=True, is_graph_task=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        harness.train_epoch(twin, labelled, loss_fn, opt, dev, classify=True, is_graph_task=True)
        torch.cuda.synchronize()
        train_ms = (time.perf_counter() - t0) / len(labelled) * 1e3
        st = harness._steps_of(twin).graph
        recs = [r["step"] for r in st.buckets.values() if r["step"] is not None] + list(st.slots.values())
        train_kernels = sorted({int(r.step.graph.kernel_nodes) for r in recs})
        captured_steps = len(recs)
    except Exception as e:
        train_ms = f"failed: {type(e).__name__}: {e}"
    ref_loop = reference_loop_leg(model, graphs[:1000], 1, True, epochs=3) if args.loop == "reference" else None
    ref_fresh = (reference_loop_leg(model, graphs[:1000], 1, True, epochs=3, fresh_inputs=True)
                 if (args.loop == "reference" and args.fresh_inputs) else None)
    ms = elapsed / args.steps * 1e3
    # the forward of a 30-node graph is one launch (small_graph_kernel) of ~20 us: a chain of latencies.  Algorithmic bytes
    # per graph: N_g^2 hop codes (1 B) + N_g x D counts + x + the weights (F MLPs of 4.3k floats)
    n_params = sum(p.numel() for p in model.parameters())
    b_alg = sum(d.x.shape[0] ** 2 + d.x.numel() * 4 for d in graphs) + len(graphs) * n_params * 4
    result = {
        "metric": "edges aggregated/sec, TensorGNAN forward (graph-level, one graph per forward)",
        "value": total_pairs / (elapsed / args.steps), "unit": "edges/s", "n_gpus": world, "ranks_seen": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"c2_mutagenicity_shaped_{args.graphs}graphs_F15_H{args.hidden}_L{args.layers}_C1_dense",
                   "graphs": args.graphs, "pairs": total_pairs, "partition": f"replicas x{world}", "exchange": "none",
                   "step": "one evaluation pass: models.TensorGNAN.forward(data) per graph, batch_size = 1"},
        "ms_per_graph": ms / max(1, len(graphs)), "graphs_per_s": args.graphs / (elapsed / args.steps),
        "replayed_eval_ms_per_graph": replay_ms, "replayed_train_ms_per_graph": train_ms,
        "kernels_per_replayed_training_step": train_kernels, "captured_training_steps": captured_steps,
        "roofline": {"bound": "hbm", "kernel": "small_graph_kernel (whole forward of a graph in one launch)",
                     "achieved": b_alg / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": b_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_launch": b_alg / max(1, len(graphs)), "avg_launch_ms": ms / max(1, len(graphs)),
                     "note": "latency-bound by construction (one ~30-node graph per launch, host-issued); bytes are not the limit"},
        "seeds": {"graphs": 0, "weights": 0}, "git_sha": git_sha(), "setup_s": t_setup,
        "timed_call": "gnan_amd.models.TensorGNAN.forward(data), eager, per graph",
        "checksum": float(sum(float(o.double().sum()) for o in outs)),
    }
    if ref_loop is not None:
        result["reference_loop"] = ref_loop
    if ref_fresh is not None:
        result["reference_loop_fresh_inputs"] = ref_fresh
    if world == 1 and not args.no_cpu_baseline:
        from oracle import gnan_oracle as O
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        p64 = {k: v.double() for k, v in sd.items()}
        torch.set_num_threads(min(os.cpu_count(), 16))
        t_cpu, cpu_pairs = 0.0, 0
        got, ref32, truth = [], [], []
        for k, (ei, x, n) in enumerate(host):
            nd, norm = O.pre_process_dense(ei, n)
            t0 = time.perf_counter()
            r32 = O.tensor_gnan_forward_models(x, nd, norm, sd, True, True, 0)
            t_cpu += time.perf_counter() - t0
            cpu_pairs += n * n
            ref32.append(r32.double().reshape(-1))
            truth.append(O.tensor_gnan_forward_models(x.double(), nd.double(), norm.double(), p64, True, True, 0).reshape(-1))
            got.append(outs[k].double().cpu().reshape(-1))
        got, ref32, truth = torch.cat(got), torch.cat(ref32), torch.cat(truth)
        # a graph's output is ONE number (a sum over nodes and features that may cancel): the sample's outputs are compared
        # as one vector, max |y - y64| / max |y64|, next to the same figure of the fp32 oracle
        worst = float((got - truth).abs().max() / truth.abs().max())
        worst32 = float((ref32 - truth).abs().max() / truth.abs().max())
        single = float(((got - truth).abs() / truth.abs().clamp_min(1e-30)).max())
        single32 = float(((ref32 - truth).abs() / truth.abs().clamp_min(1e-30)).max())
        result["cpu_baseline"] = {"value": cpu_pairs / t_cpu, "unit": "edges/s", "cores": min(os.cpu_count(), 16), "kind": "port",
                                  "cpu_model": cpu_model(),
                                  "sample": f"oracle/gnan_oracle.py (dense restatement of models.py:358-384) on the first "
                                            f"{len(host)} graphs ({cpu_pairs} pairs, {t_cpu:.2f} s), inputs from the oracle's "
                                            f"pre_process_dense",
                                  "parity_max_rel_err": worst, "cpu_f32_vs_f64_rel_err": worst32, "parity_graphs": len(host),
                                  "worst_single_graph_rel_err": single, "cpu_f32_worst_single_graph_rel_err": single32}
    print(json.dumps(result), flush=True)


def measure_traffic(kernel_prefix="spmm_kernel<"):
    """HBM bytes per launch of the dominant kernel from the PMC counters, measured NOW: two child runs of this command under
    ``rocprofv3 --kernel-trace --pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` (separate passes, as MI355X_MICROARCH.md
    prescribes; the program itself behind ``--``), 2 timed steps each.  FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 a
    128-byte memory-side request is tallied as 64 bytes by FETCH_SIZE when the kernel's loads are 16 bytes per lane over
    aligned 256-byte rows (profiles/hbm_traffic.json has the calibration), hence the factor 2 on the read side.
    Returns ``(bytes, note)`` or ``(None, why)``."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    drop = {"--traffic", "--sustain-seconds", "--steps", "--warmup"}
    argv, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a.split("=")[0] in drop:
            skip = "=" not in a
            continue
        if a != "--no-cpu-baseline":
            argv.append(a)
    argv += ["--traffic", "off", "--sustain-seconds", "0", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    found = {}
    tmp = tempfile.mkdtemp(prefix="gnan_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-f", "csv", "-d", out, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__)] + argv
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            vals = []
            for fn in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(fn)):
                    if kernel_prefix in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                return None, f"rocprofv3 --pmc {counter}: rc {r.returncode}, {len(vals)} samples"
            found[counter] = sum(vals) / len(vals)
    except Exception as e:                           # the figure is an extra: never fail the bench line for it
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = found["FETCH_SIZE"] * 1024.0 * 2.0 + found["WRITE_SIZE"] * 1024.0
    return total, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE ({found['FETCH_SIZE']:.6g} KiB x 2, gfx950 correction) + "
                   f"--pmc WRITE_SIZE ({found['WRITE_SIZE']:.6g} KiB) per launch of {kernel_prefix}...>, separate child passes")


def time_alt_partition(name, args, c):
    """``--alt-steps`` forwards of the ``vertex`` (all-gather of the whole ``[N, W]`` operand) or ``exchange`` (all-to-all-v of
    the listed halo rows) partition of the same workload, eager, stage by stage with HIP events; time = max over ranks."""
    import torch.distributed as dist
    from gnan_amd import synthetic as syn
    from gnan_amd.distributed import (VertexPartition, build_exchange_plan, halo_exchange_forward, partitioned_forward)
    N, F, C, dev, rank, world = c["N"], c["F"], c["C"], c["dev"], c["rank"], c["world"]
    if name == "vertex":                         # equal blocks: the all-gather needs equally sized shards
        part = VertexPartition(N, world, rank)
        g = syn.hop1_csr(c["src"], c["dst"], N, part.lo, part.hi)
        names = ["fmlp", "gather", "total", "spmm"]
    else:
        part = VertexPartition(N, world, rank, c["bounds"])
        xplan = build_exchange_plan(syn.hop1_csr(c["src"], c["dst"], N, part.lo, part.hi), part)
        g = xplan.halo.graph
        names = ["fmlp", "gather", "total", "spmm"]
    x = syn.block_features(N, F, part.lo, part.hi, seed=1, device=dev)
    g.long_row_plan()
    if g.n_rows >= 65536 and not g.is_dense:
        g.degree_sorted_copy()
    events = []

    def one(record):
        marks = {}

        def mark(n):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks[n] = ev
        with torch.no_grad():
            if name == "vertex":
                out = partitioned_forward(x, g, c["stacked"], c["lut"], True, part, order=args.order, out_channels=C,
                                          marks=mark, operand_dtype=c["op_dtype"])
            else:
                out = halo_exchange_forward(x, xplan, c["stacked"], c["lut"], True, order=args.order, out_channels=C,
                                            marks=mark, operand_dtype=c["op_dtype"])
        if record:
            events.append(marks)
        return out
    for _ in range(2):
        out = one(False)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.alt_steps):
        out = one(True)
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    checksum = out.double().sum().reshape(1)
    dist.all_reduce(checksum, op=dist.ReduceOp.SUM)
    stages = {n: 0.0 for n in names}
    for m in events:
        prev = m["start"]
        for n in names:
            if n in m:
                stages[n] += prev.elapsed_time(m[n]) / len(events)
                prev = m[n]
    ms = float(t) / args.alt_steps * 1e3
    W = F * C if args.order == "reference" else C
    moved = (N - (part.hi - part.lo)) * W * 4 if name == "vertex" else int(xplan.halo.halo.numel()) * W * 4
    return {"ms_per_step": ms, "edges_per_s": c["E"] / (ms / 1e3), "steps": args.alt_steps, "loop": "eager",
            "exchange": "all_gather(operand [N,W])" if name == "vertex" else "all_to_all_v(listed remote operand rows [n_halo,W])",
            "bytes_received_rank0": moved, "stages_ms_rank0": stages, "checksum": float(checksum)}


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher around it: start the N ranks ourselves.

    The parent makes no GPU call (``torch.cuda.device_count()`` does not initialise the device on this image); it runs
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py …``
    as a CHILD process (never an exec), forwards its output — rank 0 prints the one JSON line — and returns its exit code.
    Fewer visible GPUs than ranks is an error unless ``--same-device`` (dry run, all ranks on cuda:0) was asked for: the
    bench never falls through to a one-rank run labelled otherwise."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and not args.same_device:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible; refusing to time fewer ranks than asked "
              f"(dry run of the multi-rank path on one GPU: --same-device --backend gloo)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if args.config == "c3":
        if world > 1:
            raise SystemExit("config c3 (169k nodes) is a one-GPU configuration: run it with --gpus 1")
        return run_c3(args)
    if args.config == "c2":
        if world > 1:
            dist.init_process_group("gloo" if args.backend == "gloo" else "nccl",
                                    **({} if args.backend == "gloo" else {"device_id": torch.device("cuda", local_rank)}))
        run_c2(args, rank, world)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    grouped = world > 1 or args.force_dist
    if args.force_dist and world == 1:                       # no launcher set the rendezvous up: a private one
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            free = s_.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
    if grouped:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    import gnan_amd  # noqa: F401
    from gnan_amd import synthetic as syn
    from gnan_amd.distributed import (FeaturePartition, VertexPartition, build_exchange_plan, build_halo_plan,
                                      SharePipeline, balanced_bounds, choose_partition, feature_parallel_forward,
                                      halo_exchange_forward,
                                      halo_recompute_forward, partitioned_forward, slice_features)
    from gnan_amd import _lib, functional
    from gnan_amd.functional import stack_mlps
    from gnan_amd.graph import hop_inputs
    if args.index_buckets:
        functional.INDEX_BUCKETS = args.index_buckets
    apply_sets(args)
    if args.force_dist:
        from gnan_amd import distributed as _dist_mod
        _dist_mod.ALWAYS_COMMUNICATE = True
    functional.FMLP_ALGO = {"auto": _lib.FMLP_AUTO, "lane": _lib.FMLP_LANE, "mfma": _lib.FMLP_MFMA,
                            "pwl": _lib.FMLP_PWL}[args.fmlp_algo]
    from gnan_amd.models import TensorGNAN

    N, E, F, H, L, C = args.nodes, args.edges, args.feat, args.hidden, args.layers, args.out
    emulated = args.emulate_world > 1 and world == 1
    pworld = args.emulate_world if emulated else world       # how many shares the work is cut into
    args.pipeline = args.pipeline == "on" or (args.pipeline == "auto" and (pworld > 1 or args.force_dist))
    partition = args.partition if args.partition != "auto" else choose_partition(
        N, F, C, max(pworld, 2) if args.force_dist else pworld, args.order)
    if emulated and partition in ("vertex", "exchange"):
        raise SystemExit("--emulate-world covers the halo and feature partitions (these shares need the other ranks' operand rows)")
    if partition == "feature" and args.order != "reference":
        raise SystemExit("--partition feature needs --order reference (sum-first exchanges the narrow operand)")
    t_setup = time.perf_counter()
    src, dst = syn.rmat_edges(args.scale, N, E, seed=0, device=dev)
    bounds = None
    if pworld > 1 and partition in ("halo", "exchange") and args.cut == "cost":
        # every rank draws the same edges, hence the same degrees and the same cut
        bounds = balanced_bounds(torch.bincount(src, minlength=N) + 1, pworld)
    part = VertexPartition(N, pworld, rank, bounds)
    fpart = FeaturePartition(F, pworld, rank)
    if emulated:                                             # same shares, no process group: collectives are skipped
        part.world = fpart.world = 1
        if bounds is None:
            part.__class__ = type("EmuV", (VertexPartition,), {"block": property(lambda self: -(-N // pworld))})
        fpart.__class__ = type("EmuF", (FeaturePartition,), {"block": property(lambda self: -(-F // pworld))})
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
    want_alt = grouped and not emulated and args.alt_partitions != "off" and partition == "halo" and args.order == "reference"
    if not want_alt:
        del src, dst
    torch.manual_seed(0)
    model = TensorGNAN(F, C, L, hidden_channels=H, normalize_rho=True, rho_per_feature=False, device="cuda")
    with torch.no_grad():                                    # O(1)-scale weights (the upstream init gives ~1e-14 outputs)
        for _, p in model.named_parameters():
            if p.dim() == 2:
                torch.nn.init.xavier_normal_(p, gain=1.0)
            else:
                p.normal_(0.0, 0.5)
    model = model.to(dev).eval()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    # What the timed step does NOT contain because it is derived from the INPUTS once and kept (DESIGN.md sections 3 / 4.2): the
    # degree-sorted copy of the CSR with its hub-row plan, and the per-feature value range of x the direct-index look-up is laid
    # over.  Built here explicitly (they are cached from now on) so that their cost is in the line.
    torch.cuda.synchronize()
    amortised = {}
    t0 = time.perf_counter()
    g.long_row_plan()
    if g.n_rows >= 65536 and not g.is_dense:
        g.degree_sorted_copy()
    torch.cuda.synchronize()
    amortised["degree_sorted_csr_copy_and_hub_plan_ms"] = (time.perf_counter() - t0) * 1e3
    if args.order == "sum_first" and partition == "vertex":
        # the narrow aggregation walks a bucketed copy of the pairs (HopGraph.pb_plan -> gnan_spmm_pb_fwd): one sort of the pairs
        from gnan_amd import aggregate as _agg
        t0 = time.perf_counter()
        if _agg.PB_NARROW and _agg.PB_MIN_NNZ <= g.nnz <= _agg.PB_MAX_NNZ:
            g.pb_plan(C)
        torch.cuda.synchronize()
        amortised["bucketed_pairs_plan_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    functional._feature_range(x)
    torch.cuda.synchronize()
    amortised["feature_range_pass_ms"] = (time.perf_counter() - t0) * 1e3
    amortised["total_ms"] = sum(amortised.values())

    with torch.no_grad():
        stacked = stack_mlps(model.fs)
        lut = model.rho(hop_inputs(g.n_codes, dev).view(-1, 1))
    op_dtype = torch.bfloat16 if args.operand == "bf16" else torch.float32
    stage_names = ["fmlp", "gather", "total", "spmm"] + (["reduce"] if partition == "feature" else [])
    # One GPU: the timed call is the drop-in itself, TensorGNAN.forward(data) (models.py:358-384) on a data object that
    # carries the hop-coded graph — graph lookup, weight views, rho on the D distinct distances, look-up, aggregation.
    # More than one rank: gnan_amd.distributed's forward of this rank's share (the reference has no multi-process path).
    use_module = world == 1 and not emulated and partition == "vertex" and not args.pipeline and not args.force_dist
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
    # ... and the whole share as one graph launch per forward (halo / vertex partitions, inference)
    share, share_note = None, None
    if grouped and args.backend != "nccl" and args.share_graph != "off":
        share_note = f"{args.backend} collectives go through the host and cannot be captured: eager loop"
    elif args.share_graph != "off" and prefetch is not None and (args.share_graph == "on" or args.pipeline):
        from gnan_amd.graphed import CaptureFailed

        def share_forward(tables, marks=None):
            if partition == "halo":
                return halo_recompute_forward(x, plan, stacked, lut, True, order=args.order, out_channels=C,
                                              operand_dtype=op_dtype, tables=tables, marks=marks)
            return partitioned_forward(x, g, stacked, lut, True, part, order=args.order, out_channels=C,
                                       operand_dtype=op_dtype, tables=tables, marks=marks)
        def all_ranks(ok: bool) -> bool:             # every rank replays, or none does (the collectives must pair up)
            if not grouped:
                return ok
            flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(int(flag))
        want = None
        try:
            with torch.no_grad():
                want = share_forward(prefetch.launch()).clone()
            share = SharePipeline(share_forward, stacked, x=x, fork_at=args.share_fork)
        except (CaptureFailed, RuntimeError) as e:
            share, share_note = None, f"not capturable ({type(e).__name__}: {str(e)[:200]}): eager loop"
        if not all_ranks(share is not None):
            if share is not None:
                share_note = "another rank could not capture its share: eager loop on every rank"
            share = None
        if share is not None:                        # (same number of replays — of captured collectives — on every rank)
            got = [share.step().clone() for _ in range(4)]          # both graphs twice: a node that replays correctly only once shows
            torch.cuda.synchronize()
            scale = float(want.abs().max())
            worst = max(float((o - want).abs().max()) for o in got)
            good = not share.tripped() and worst <= 1e-6 * scale
            if not all_ranks(good):
                share = None
                share_note = (f"first replays off by {worst / max(scale, 1e-30):.2e} (or guard tripped): eager loop" if not good
                              else "another rank's replays did not reproduce its eager forward: eager loop on every rank")

    def step(record):
        marks = {}
        if record:
            def mark(name):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks[name] = ev
        else:
            mark = None
        if share is not None:                       # one graph launch: look-up, collective, aggregation + the next table build
            if record:
                mark("start")
            out = share.step()
            if record:
                mark("spmm")
                events.append(marks)
            return out
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
        if grouped:
            dist.barrier()

    # The warm-up steps are the timed steps to the letter — events recorded, the previous step's output still held while the
    # next one runs — so that the caching allocator has reached the timed loop's steady state before the clock starts: with
    # `step(False)` and the result dropped, the first TIMED step could find its 2.56-GB operand block carved up for the kept
    # output and ask the driver for a fresh one (hipMalloc of 2.56 GB: 0.4 s on a loaded host, i.e. +20 ms on each of 20 steps).
    out = None
    for _ in range(args.warmup):
        out = step(True)
    events.clear()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = None
    if grouped:                                      # every rank's own time (one all-gather); the line's time is the slowest
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = torch.empty(dist.get_world_size(), device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, t)
        every = (every / args.steps * 1e3).tolist()
        rank_ms = {"min": min(every), "max": max(every), "per_rank": every}
        elapsed = max(every) * args.steps / 1e3

    checksum = out.double().sum().reshape(1)
    if grouped and partition != "feature":      # row-partitioned output: add the ranks' shares (feature: already whole)
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
    committed = None
    if os.path.exists(tpath):
        rec = json.load(open(tpath))
        if rec.get("workload") == workload and rec.get("n_gpus") == world:
            committed = rec.get("bytes_per_launch")
    if args.traffic == "measure" and world == 1 and not emulated and rank == 0:
        traffic, traffic_source = measure_traffic()
        if traffic is None and committed is not None:
            traffic, traffic_source = committed, f"profiles/hbm_traffic.json (committed figure; live measurement failed: {traffic_source})"
    elif args.traffic != "off" and committed is not None:
        # PMC counters need rocprofv3 around the process: the figure is the committed one of the same command, not of this run
        traffic, traffic_source = committed, "profiles/hbm_traffic.json (%s; rocprofv3 --pmc passes of this command, not measured in this run)" % rec.get("source", "committed")

    result = None
    if rank == 0 or emulated:                        # (an emulated share is its own one-process job whatever RANK selects)
        ms = elapsed / args.steps * 1e3
        result = {
            "metric": "edges aggregated/sec, TensorGNAN forward", "value": E / (elapsed / args.steps),
            "unit": "edges/s", "n_gpus": world, "ranks_seen": dist.get_world_size() if grouped else 1,
            "process_group": dist.get_backend() if grouped else None, "collectives_forced": bool(args.force_dist) or None,
            "rank_ms_per_step": rank_ms, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.operand == "f32" else "bf16 operand storage, f32 accumulate",
            "data": "synthetic",
            "config": {"workload": workload, "nodes": N, "edges": E, "stored_pairs_rank0": g.nnz,
                       "operand_width": W, "partition": f"{partition} x{world}", "exchange":
                       "none" if not grouped else ("all_reduce(column sums [W])" if partition == "halo" else
                                                  "all_gather(operand [N,W])" if partition == "vertex" else
                                                  "all_to_all_v(listed remote operand rows [n_halo,W]) + all_reduce(column sums [W])"
                                                  if partition == "exchange" else "all_reduce(out [N,C])")},
            "roofline": {"bound": "hbm", "kernel": "spmm_kernel" if W * (2 if args.operand == "bf16" else 4) >= 32 else
                         "pb_expand_kernel + pb_reduce_kernel (or spmm_hot_kernel)", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_source, "traffic_over_algorithmic": traffic / b_alg if traffic else None,
                         "algorithmic_bytes_per_launch": b_alg, "avg_launch_ms": stages["spmm"],
                         # narrow operand rows (sum-first: W * 4 < 128 B): every stored pair is one sub-line gather; bytes are
                         # the wrong yardstick there (the kernel is bound by gather requests: L2 hits, LDS-resident hub rows
                         # and misses at ~55 G/s), so the gather rate is stated next to the byte fraction
                         "gathers_per_launch": int(g.nnz) if W * (2 if args.operand == "bf16" else 4) < 128 else None,
                         "gathers_G_per_s": (g.nnz / spmm_s / 1e9) if (W * (2 if args.operand == "bf16" else 4) < 128 and spmm_s > 0) else None},
            "stages_ms": stages,
            "step_ms_device": {"min": per_step[0], "median": per_step[len(per_step) // 2]} if per_step else None,
            "seeds": {"graph": 0, "features": 1, "weights": 0}, "git_sha": git_sha(),
            "spmm_edges_per_s": (g.nnz - g.n_rows) / spmm_s if spmm_s > 0 else None,
            "fmlp": fmlp_stage(args, x, H, L, C, W if args.order == "reference" else C, stages["fmlp"]),
            "pipelined_table_build": prefetch is not None,
            "share_replayed_from_hipgraphs": share is not None, "share_graph_note": share_note,
            "share_guard_tripped": share.tripped() if share is not None else None,
            "timed_call": "gnan_amd.models.TensorGNAN.forward(data)" if use_module else f"gnan_amd.distributed ({partition})",
            "emulated_share_of": pworld if emulated else None, "setup_s": t_setup, "checksum": checksum,
            "amortised_setup_ms": amortised,
            "operand_rows_rank0": int(x.shape[0]),
            "cut": None if pworld == 1 else ("cost" if bounds is not None else "rows"),
            "owned_rows_rank0": [part.lo, part.hi] if partition != "feature" else None,
            "alt_partitions": None,
            # bf16 operand rows are an inference format: the library has no backward through them
            "backward": False if args.operand == "bf16" else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            with torch.no_grad():
                from gnan_amd.functional import feature_mlps
                operand = feature_mlps(x, stacked, args.order == "sum_first")
            result["cpu_baseline"] = cpu_baseline(args, model, g, x, operand, out)
        if world == 1 and not emulated and args.sustain_seconds > 0:
            # the same step, back to back, long enough for an outside observer of the GPU to see it
            torch.cuda.synchronize()
            t0, n_sus = time.perf_counter(), 0
            while time.perf_counter() - t0 < args.sustain_seconds:
                for _ in range(50):
                    step(False)
                n_sus += 50
                torch.cuda.synchronize()
            result["sustained_ms_per_step"] = (time.perf_counter() - t0) / n_sus * 1e3
            result["sustained_steps"] = n_sus
    # ---- the partitions that move operand rows between the ranks, a few steps each (north_star's halo all-gather over xGMI
    #      measured next to the halo-recompute line; every rank takes part, rank 0 prints) ----------------------------------
    if want_alt:
        if result is not None:           # should a collective of the extra partitions hang, the measured line is at least in the log
            print("bench.py: the line so far (alt_partitions follow): " + json.dumps(result), file=sys.stderr, flush=True)
        alt = {}

        def bail():
            # a collective of the extra partitions that never returns must not cost the measured line: every rank runs this
            # timer (started behind the same barrier), rank 0 prints the line with what was finished, all leave with 0
            if result is not None:
                for name in ("vertex", "exchange"):
                    alt.setdefault(name, {"error": f"not finished within --alt-timeout {args.alt_timeout} s"})
                result["alt_partitions"] = alt
                print(json.dumps(result), flush=True)
            sys.stderr.flush()
            os._exit(0)
        import threading
        timer = threading.Timer(args.alt_timeout, bail)
        timer.daemon = True
        timer.start()
        for name in ("vertex", "exchange"):
            try:
                alt[name] = time_alt_partition(name, args, dict(N=N, E=E, F=F, C=C, dev=dev, rank=rank, world=world, src=src, dst=dst,
                                                                stacked=stacked, lut=lut, op_dtype=op_dtype, bounds=bounds))
            except Exception as e:                   # (every rank fails alike or the next collective hangs: shapes are the same on all)
                alt[name] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        timer.cancel()
        del src, dst
        if result is not None:
            result["alt_partitions"] = alt
    if result is not None:
        print(json.dumps(result), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
