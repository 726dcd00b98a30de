# Same-box A/B of two builds of the library: GNAN_HIP_LIB selects the shared object (see gnan_amd/_lib.py).
# usage: bash tools/ab_lib.sh <lib_a.so> <lib_b.so> [bench flags]
A=$1; B=$2; shift 2
for rep in 1 2; do for lib in "$A" "$B"; do for op in f32 bf16; do
  GNAN_HIP_LIB=$PWD/$lib python bench.py --traffic committed --sustain-seconds 0 --steps 10 --warmup 3 --no-cpu-baseline --operand $op "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', '$op', {k: round(v, 3) for k, v in d['stages_ms'].items()}, d['checksum'])"
done; done; done
