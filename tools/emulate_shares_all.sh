# Every one of the P shares of a P-rank halo-recompute job, timed one after the other on ONE GPU (RANK=r selects the share).
# FORCE=1: each share also creates a one-rank process group on RCCL and runs its collectives over it (--force-dist): the
# all-reduce captured in the share's hipGraphs is then a real RCCL launch.  Prints one line per share and the slowest.
P=${P:-8}
EXTRA=""
[ "${FORCE:-0}" = "1" ] && EXTRA="--force-dist"
for r in $(seq 0 $((P-1))); do
  RANK=$r python bench.py --traffic off --sustain-seconds 0 --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --emulate-world $P --partition halo $EXTRA "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('share', $r, 'of', $P, 'wall_ms', round(d['ms_per_step'],4), 'device', d['step_ms_device'], 'rows', d['operand_rows_rank0'], 'pairs', d['config']['stored_pairs_rank0'], 'graphs' if d['share_replayed_from_hipgraphs'] else 'eager', d['share_graph_note'] or '', 'group', d.get('process_group'), 'checksum', d['checksum'])"
done | tee /tmp/shares_$$.txt
python - <<PY
rows=[l.split() for l in open('/tmp/shares_$$.txt') if l.startswith('share')]
w=[float(r[5]) for r in rows]
print('slowest share wall_ms', max(w), 'fastest', min(w), 'mean', sum(w)/len(w))
PY
