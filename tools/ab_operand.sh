for op in f32 bf16; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --operand $op 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$op', d['stages_ms'], d['checksum'])"
done
