# Per-rank stage times of a P-rank halo-recompute job, measured on ONE GPU (collectives excluded): P = 1, 2, 4, 8.
# Shares (P > 1) are replayed from hipGraphs (distributed.SharePipeline); SHARE_GRAPH=off gives the eager loop.
for P in 1 2 4 8; do
  python bench.py --traffic committed --sustain-seconds 0 --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline --emulate-world $P --partition halo --share-graph ${SHARE_GRAPH:-auto} "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print($P, 'wall', round(d['ms_per_step'],3), 'device', d['step_ms_device'], {k: round(v, 3) for k, v in d['stages_ms'].items()}, d['operand_rows_rank0'], d['config']['stored_pairs_rank0'], 'graphs' if d['share_replayed_from_hipgraphs'] else 'eager', d['share_graph_note'] or '', 'checksum', d['checksum'])"
done
