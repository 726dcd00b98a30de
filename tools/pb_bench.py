"""Narrow (W = 1, 2) aggregation on the C4 graph: propagation-blocked kernels (csrc/spmm_pb.hip) against the row-parallel
ones (spmm_hot_kernel), same box, alternating; plan construction time; parity of the two.   python tools/pb_bench.py [W ...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnan_amd  # noqa
from gnan_amd import aggregate, synthetic as syn
from gnan_amd.aggregate import spmm_launch

dev = torch.device("cuda")
N, E = 10_000_000, 100_000_000
src, dst = syn.rmat_edges(24, N, E, seed=0, device=dev)
g = syn.hop1_csr(src, dst, N)
del src, dst
lut = torch.tensor([[0.7], [-0.3], [0.2]], device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, y


for W in [int(w) for w in sys.argv[1:]] or [1, 2]:
    S = torch.rand((N, W), device=dev)
    total = S.sum(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan = g.pb_plan(W)
    torch.cuda.synchronize()
    t_plan = (time.perf_counter() - t0) * 1e3
    out = {"W": W, "plan_ms": t_plan, "entries": plan.n_entries, "pairs": plan.n_pairs, "bins": plan.n_bins,
           "column_blocks": plan.n_cblocks, "chunks": int(plan.chunk_q.numel())}
    flag_sets = [int(f, 0) for f in os.environ.get("PB_FLAG_SETS", "0").split(",")]
    USE_CNT, WITH_REST = os.environ.get("PB_USE_CNT", "1") == "1", os.environ.get("PB_WITH_REST", "1") == "1"
    for rnd in range(2):
        for fl in flag_sets:
            aggregate.PB_NARROW, aggregate.PB_FLAGS = True, fl
            ms_pb, y_pb = timed(lambda: spmm_launch(g, S, lut, USE_CNT, WITH_REST, s_total=total if WITH_REST else None))
            out[f"pb_ms_flags{fl:#x}_{rnd}"] = ms_pb
        aggregate.PB_NARROW, aggregate.PB_FLAGS = False, 0
        ms_rows, y_rows = timed(lambda: spmm_launch(g, S, lut, USE_CNT, WITH_REST, s_total=total if WITH_REST else None))
        out[f"rows_ms_{rnd}"] = ms_rows
    out["max_rel_diff"] = float((y_pb - y_rows).abs().max() / y_rows.abs().max())
    print(json.dumps(out), flush=True)
