#!/usr/bin/env python3
"""Everything `bash tools/profile_round.sh <tag>` left under gpurun_out/<tag>, condensed into profiles/<tag>_* :

    python profiles/condense_round.py r05

(profiles/summarize.py for the four rocprofv3 passes of the default bench; the JSON line of every bench log; per-kernel
statistics of the other profiled commands; the text outputs as they are.)"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(here), "gpurun_out", tag)
pre = os.path.join(here, tag)
ours = ("spmm", "fmlp", "fpwl", "pwl_", "dense_to_code", "colsum", "bfs_", "loss_", "small_graph", "dense_lut", "multi_copy",
        "rho_row", "gather_rows", "absmax", "scales_kernel", "pack_bwd", "index_build", "sum_groups")


def last_json(path):
    lines = [l for l in open(path) if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def json_of(log, out):
    p = os.path.join(src, log)
    if os.path.exists(p):
        d = last_json(p)
        if d is not None:
            json.dump(d, open(pre + out, "w"))
            open(pre + out, "a").write("\n")
            return d
    print("missing", log)


def stats_of(sub, out):
    found = glob.glob(os.path.join(src, sub, "**", "*_kernel_stats.csv"), recursive=True)
    if not found:
        print("missing", sub)
        return
    rows = list(csv.DictReader(open(found[0])))
    with open(pre + out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for n, r in enumerate(rows):
            if n < 25 or any(o in r["Name"] for o in ours):
                w.writerow(r)


def text_of(name, out):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copyfile(p, pre + out)
    else:
        print("missing", name)


subprocess.check_call([sys.executable, os.path.join(here, "summarize.py"), src, pre])
for log, out in (("bench_full.log", "_bench_full.json"), ("sum_first_bench.log", "_sum_first_bench.json"),
                 ("train_step_noprof.log", "_train_step_c4.json"), ("train_step_per_layer.log", "_train_step_c4_per_layer_optimizer.json"), ("c5_bench.log", "_c5_bench.json"),
                 ("c5_sum_first_bench.log", "_c5_sum_first_bench.json"), ("c3_bench.log", "_c3_bench.json"),
                 ("c3_c40_bench.log", "_c3_c40_bench.json"), ("c2_bench.log", "_c2_bench.json"),
                 ("c3_loop_reference.log", "_c3_loop_reference.json"), ("c2_loop_reference.log", "_c2_loop_reference.json"),
                 ("bench_force_dist_rccl.log", "_bench_force_dist_rccl.json"), ("muta_epoch.json", "_muta_epoch.json"),
                 ("c3_loop_reference_fresh.log", "_c3_loop_reference_fresh.json"),
                 ("bench_force_dist_alt_partitions.log", "_bench_force_dist_alt_partitions.json"), ("c1_cpu.json", "_c1_cpu_on_the_box.json")):
    json_of(log, out)
stats_of("sum_first", "_sum_first_kernel_stats.csv")
stats_of("train", "_train_step_c4_kernel_stats.csv")
for name, out in (("emulated_shares.txt", "_emulated_shares.txt"), ("emulated_shares_all.txt", "_emulated_shares_all.txt"),
                  ("emulated_shares_all_rccl.txt", "_emulated_shares_all_rccl.txt"), ("lookup_ab.jsonl", "_lookup_ab.jsonl"),
                  ("sq_fwd.txt", "_sq_fwd.txt"), ("sq_train.txt", "_sq_train.txt"), ("batched_bench.jsonl", "_batched_bench.jsonl"),
                  ("timeline_muta.txt", "_timeline_muta.txt"), ("c3_timeline.txt", "_c3_timeline.txt"), ("c3_c40_timeline.txt", "_c3_c40_timeline.txt"),
                  ("reference_loop.jsonl", "_reference_loop.jsonl"), ("reference_loop_host_profile.txt", "_reference_loop_host_profile.txt"),
                  ("share_timeline.txt", "_share_timeline.txt"), ("emulated_shares_all_cut_rows.txt", "_emulated_shares_all_cut_rows.txt"),
                  ("pb_bench.jsonl", "_pb_bench.jsonl"), ("pmc_narrow.csv", "_pmc_narrow.csv"),
                  ("reference_loop_faces.jsonl", "_reference_loop_faces.jsonl")):
    text_of(name, out)
p = os.path.join(src, "graphed_steps.log")
if os.path.exists(p):
    open(pre + "_graphed_steps.jsonl", "w").writelines(l for l in open(p) if l.startswith("{"))
p = os.path.join(src, "gpu_suite_durations.txt")
if os.path.exists(p):
    lines = open(p).read().splitlines()
    keep = [l for l in lines if "slowest" in l or "s call" in l or "s setup" in l or "passed" in l or "failed" in l]
    open(pre + "_gpu_suite_durations.txt", "w").write("\n".join(keep) + "\n")
print("condensed", src, "->", pre + "_*")
