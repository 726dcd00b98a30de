# Every measurement behind profiles/<tag>_* in one go, on the GPU box:   bash tools/profile_round.sh r03
#   default bench (4 rocprofv3 passes), sum-first forward, training steps (sum-first and reference order), config 5 on one
#   GPU (both orders), emulated per-rank shares.  Condense with  python profiles/summarize.py gpurun_out/<tag> profiles/<tag>
TAG=${1:-r03}
OUT=gpurun_out/$TAG
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $OUT
bash tools/profile_bench.sh $OUT > $OUT/profile_bench.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/sum_first -o sf -- python3 bench.py --traffic committed --sustain-seconds 0 --order sum_first --no-cpu-baseline --steps 10 --warmup 3 > $OUT/sum_first_bench.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/train -o tr -- python3 tools/train_step_c4.py > $OUT/train_step.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/train_ref -o trr -- python3 tools/train_step_c4_reference_order.py > $OUT/train_step_reference.log 2>&1
python3 tools/train_step_c4.py > $OUT/train_step_noprof.log 2>&1
python3 bench.py --traffic off --sustain-seconds 0 --scale 27 --nodes 111059956 --edges 1615685872 --operand bf16 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/c5_bench.log 2>&1
python3 bench.py --traffic off --sustain-seconds 0 --scale 27 --nodes 111059956 --edges 1615685872 --order sum_first --no-cpu-baseline --steps 5 --warmup 2 > $OUT/c5_sum_first_bench.log 2>&1
bash tools/emulate_shares.sh > $OUT/emulated_shares.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_full.log 2>&1
python3 tools/graphed_step.py arxiv cora muta arxiv40 > $OUT/graphed_steps.log 2>&1          # harness epochs, eager vs replayed
python3 tools/small_graph_bench.py > $OUT/small_graph.log 2>&1
bash tools/graphed_timeline.sh $OUT/tl_muta muta > $OUT/timeline_muta.txt 2>&1                 # one replayed graph-task training step, kernel by kernel
bash tools/graphed_timeline.sh $OUT/tl_arxiv arxiv > $OUT/timeline_arxiv.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
du -sh $OUT
