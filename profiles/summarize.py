#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<round>/...) into the small files committed under profiles/.

    python profiles/summarize.py gpurun_out/r01 profiles/r01

writes  <prefix>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats summary, top 25 kernels)
        <prefix>_pmc_summary.csv    (per kernel, per counter: launches, mean value per launch)
        <prefix>_bench.json         (the bench line printed under the profiler)
HBM bytes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide (16 B/lane) coalesced
reads by up to 2x (MI355X_MICROARCH.md §HBM), so read traffic is reported as a [1x, 2x] interval.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, prefix = sys.argv[1], sys.argv[2]
ours = ("spmm", "fmlp", "fpwl", "pwl_", "dense_to_code", "colsum", "bfs_", "loss_", "small_graph", "dense_lut", "multi_copy", "rho_row", "gather_rows", "absmax", "scales_kernel", "pack_bwd")

rows = list(csv.DictReader(open(glob.glob(os.path.join(src, "stats", "*_kernel_stats.csv"))[0])))
with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for n, r in enumerate(rows):                      # top 25 of the process + every kernel of this library
        if n < 25 or any(o in r["Name"] for o in ours):
            w.writerow(r)

acc = collections.defaultdict(list)
for d in ("fetch", "write", "tcc"):
    for fn in glob.glob(os.path.join(src, d, "*_counter_collection.csv")):
        for r in csv.DictReader(open(fn)):
            if any(o in r["Kernel_Name"] for o in ours):
                m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
                acc[(m.group(1) if m else r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(prefix + "_pmc_summary.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "mean_per_launch", "note"])
    for (k, c), v in sorted(acc.items()):
        m = sum(v) / len(v)
        note = ""
        if c == "FETCH_SIZE":
            note = f"HBM read bytes/launch in [{m * 1024:.4g}, {2 * m * 1024:.4g}] (KiB counter; gfx950 1x-2x correction)"
        if c == "WRITE_SIZE":
            note = f"HBM write bytes/launch ~ {m * 1024:.4g}"
        w.writerow([k, c, len(v), f"{m:.6g}", note])

for line in open(os.path.join(src, "stats_bench.log")):
    if line.startswith("{"):
        json.dump(json.loads(line), open(prefix + "_bench.json", "w"), indent=1)
print("wrote", prefix + "_*")
