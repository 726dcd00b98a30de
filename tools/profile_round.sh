# Every measurement behind profiles/<tag>_* in one go, on the GPU box:   bash tools/profile_round.sh r04
#   default bench (4 rocprofv3 passes), sum-first forward, training steps (sum-first and reference order), config 5 on one
#   GPU (both orders), configs 2 and 3 (bench.py --config), emulated per-rank shares (replayed from hipGraphs and eager),
#   look-up A/B + the access-pattern ceiling + SQ counters, batched graphs, harness epochs; round 5: config 3 timeline, the
#   reference-shaped loop, RCCL on a one-rank group, all eight shares, the GPU suite's durations.
#   Condense with  python profiles/summarize.py gpurun_out/<tag> profiles/<tag>
TAG=${1:-r04}
OUT=gpurun_out/$TAG
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $OUT
bash tools/profile_bench.sh $OUT > $OUT/profile_bench.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/sum_first -o sf -- python3 bench.py --traffic off --sustain-seconds 0 --order sum_first --no-cpu-baseline --steps 10 --warmup 3 > $OUT/sum_first_bench.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/train -o tr -- python3 tools/train_step_c4.py > $OUT/train_step.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/train_ref -o trr -- python3 tools/train_step_c4_reference_order.py > $OUT/train_step_reference.log 2>&1
python3 tools/train_step_c4.py > $OUT/train_step_noprof.log 2>&1
PER_LAYER=1 python3 tools/train_step_c4.py > $OUT/train_step_per_layer.log 2>&1              # the optimizer over model.parameters() (torch's per-tensor overhead on top)
python3 bench.py --config c5 --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/c5_bench.log 2>&1
python3 bench.py --config c5 --operand f32 --order sum_first --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/c5_sum_first_bench.log 2>&1
python3 bench.py --config c3 --steps 20 --warmup 5 > $OUT/c3_bench.log 2>&1
python3 bench.py --config c3 --out 40 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/c3_c40_bench.log 2>&1
python3 bench.py --config c2 --steps 3 --warmup 1 > $OUT/c2_bench.log 2>&1
bash tools/emulate_shares.sh > $OUT/emulated_shares.txt 2>&1
SHARE_GRAPH=off bash tools/emulate_shares.sh > $OUT/emulated_shares_eager.txt 2>&1
python3 bench.py > $OUT/bench_full.log 2>&1                                                   # the driver's command: traffic measured, CPU baseline, sustained leg
python3 tools/lookup_ab.py > $OUT/lookup_ab.jsonl 2> $OUT/lookup_ab.err
tools/_bin/lookup_ceiling > $OUT/lookup_ceiling.jsonl 2>&1
bash tools/pmc_sq_cmd.sh $OUT/sq_fwd python3 bench.py --traffic off --sustain-seconds 0 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq_fwd.txt 2>&1
bash tools/pmc_sq_cmd.sh $OUT/sq_train python3 tools/train_step_c4.py > $OUT/sq_train.txt 2>&1
python3 tools/batched_bench.py > $OUT/batched_bench.jsonl 2> $OUT/batched_bench.err
python3 tools/graphed_step.py arxiv cora muta arxiv40 > $OUT/graphed_steps.log 2>&1          # harness epochs, eager vs replayed
python3 tools/small_graph_bench.py > $OUT/small_graph.log 2>&1
python3 tools/muta_epoch.py > $OUT/muta_epoch.json 2> $OUT/muta_epoch.err                   # config 2 at its own size: 4337 graphs per epoch
bash tools/graphed_timeline.sh $OUT/tl_muta muta > $OUT/timeline_muta.txt 2>&1                 # one replayed graph-task training step, kernel by kernel
# ---- round 5 additions
bash tools/c3_timeline.sh > $OUT/c3_timeline.txt 2>&1                                          # one replayed forward + backward step of config 3, kernel by kernel
bash tools/c3_timeline.sh --out 40 > $OUT/c3_c40_timeline.txt 2>&1                              # ... with 40 output channels (the ogbn-arxiv task's width)
python3 tools/reference_loop_bench.py --profile > $OUT/reference_loop.jsonl 2> $OUT/reference_loop_host_profile.txt   # the reference-SHAPED loop (anomaly mode, stock Adam)
python3 bench.py --config c3 --loop reference --no-cpu-baseline > $OUT/c3_loop_reference.log 2>&1
python3 bench.py --config c2 --loop reference --no-cpu-baseline --steps 3 --warmup 1 > $OUT/c2_loop_reference.log 2>&1
python3 bench.py --force-dist --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 50 > $OUT/bench_force_dist_rccl.log 2>&1   # one rank, RCCL group, collectives on
P=8 bash tools/emulate_shares_all.sh > $OUT/emulated_shares_all.txt 2>&1                      # every one of the 8 shares
FORCE=1 P=8 bash tools/emulate_shares_all.sh > $OUT/emulated_shares_all_rccl.txt 2>&1         # ... each over a one-rank RCCL group (captured all-reduce)
python3 -m pytest tests -q -m gpu --durations=15 > $OUT/gpu_suite_durations.txt 2>&1          # the whole GPU suite: total time and the slowest tests
# ---- round 6 additions
RANK=2 bash tools/step_timeline.sh $OUT/share_tl --emulate-world 8 --partition halo --force-dist > $OUT/share_timeline.txt 2>&1   # the slowest 1/8 share, kernel by kernel
P=8 bash tools/emulate_shares_all.sh --cut rows > $OUT/emulated_shares_all_cut_rows.txt 2>&1                                   # ... with blocks of equal row count, for comparison
python3 tools/pb_bench.py 1 2 > $OUT/pb_bench.jsonl 2> $OUT/pb_bench.err                                                       # propagation-blocked narrow aggregation against the row-parallel kernels
bash tools/pmc_narrow.sh $OUT > $OUT/pmc_narrow.log 2>&1                                                                       # L2 hit / miss / fabric requests of the narrow kernels, old and new
for f in "" "--flat"; do python3 tools/reference_loop_bench.py arxiv cora muta $f 2>/dev/null | grep -v "by phase"; python3 tools/reference_loop_bench.py arxiv cora muta --fresh-inputs $f 2>/dev/null; done > $OUT/reference_loop_faces.jsonl
python3 bench.py --config c3 --loop reference --fresh-inputs --no-cpu-baseline > $OUT/c3_loop_reference_fresh.log 2>&1
python3 bench.py --force-dist --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 20 --partition halo > $OUT/bench_force_dist_alt_partitions.log 2>&1   # one rank, RCCL group: alt_partitions code path
python3 tools/c1_cpu.py 3 > $OUT/c1_cpu.json 2> $OUT/c1_cpu.err                                                                # config 1 on the box's host cores
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
du -sh $OUT
