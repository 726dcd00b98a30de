# Same-box A/B of module-level constants: bash tools/ab_env.sh "aggregate.HOT_ROWS_IN_LDS=False" "aggregate.HOT_ROWS_IN_LDS=True" ... -- [bench flags]
SETS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETS+=("$1"); shift; done; shift
for rep in 1 2; do for s in "${SETS[@]}"; do
  python bench.py --traffic off --sustain-seconds 0 --steps 10 --warmup 3 --no-cpu-baseline --set "$s" "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$s', {k: round(v, 3) for k, v in d['stages_ms'].items()}, d['checksum'])"
done; done
