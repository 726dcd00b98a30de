# Same-box A/B of environment settings: bash tools/ab_env.sh "VAR=a" "VAR=b" ... -- [bench flags]
SETS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETS+=("$1"); shift; done; shift
for rep in 1 2; do for s in "${SETS[@]}"; do
  env $s python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$s', {k: round(v, 3) for k, v in d['stages_ms'].items()}, d['checksum'])"
done; done
