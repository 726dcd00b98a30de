# rocprofv3 evidence for the default bench (run on the GPU box):  bash tools/profile_bench.sh gpurun_out/r01b
# pass 1: --kernel-trace --stats of `bench.py --steps 10 --warmup 3`; passes 2-4: one PMC counter set each
# (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum) with --kernel-trace only, as gpurun requires.
# Condense with:  python profiles/summarize.py gpurun_out/r01b profiles/r01b
OUT=${1:-gpurun_out/prof}; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- python3 bench.py --traffic committed --sustain-seconds 0 --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/stats_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o fetch -- python3 bench.py --traffic committed --sustain-seconds 0 --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o write -- python3 bench.py --traffic committed --sustain-seconds 0 --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/tcc -o tcc -- python3 bench.py --traffic committed --sustain-seconds 0 --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/tcc.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +4M -delete   # keep the merged scratch small
ls -R $OUT | head -40
