# fp32 and bf16 aggregation across builds of the library: bash tools/ab_libs.sh lib1.so lib2.so ...
for rep in 1 2; do for lib in "$@"; do for op in f32 bf16; do
  GNAN_HIP_LIB=$PWD/$lib python bench.py --traffic committed --sustain-seconds 0 --steps 10 --warmup 3 --no-cpu-baseline --operand $op 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib'.split('/')[-1], '$op', round(d['stages_ms']['spmm'], 3), d['checksum'])"
done; done; done
