OUT=gpurun_out/r06c; mkdir -p $OUT
share() { RANK=2 python bench.py --traffic off --sustain-seconds 0 --steps 200 --warmup 10 --no-cpu-baseline --emulate-world 8 --partition halo --force-dist --share-fork fmlp "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wall_ms', round(d['ms_per_step'],4), 'device', d['step_ms_device'], 'rows', d['operand_rows_rank0'], 'pairs', d['config']['stored_pairs_rank0'], 'graphs' if d['share_replayed_from_hipgraphs'] else 'eager', d['share_graph_note'] or '', 'checksum', d['checksum'])"; }
echo "== share longest-first"; share
echo "== share shortest-first"; share --set graph.DEGREE_ORDER_LONGEST_FIRST=False
for o in "" "--set graph.DEGREE_ORDER_LONGEST_FIRST=False"; do
 echo "== N=1 reference order $o"; python bench.py --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 20 $o 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms', round(d['ms_per_step'],4), d['stages_ms'], d['checksum'])"
 echo "== N=1 sum_first $o"; python bench.py --traffic off --sustain-seconds 0 --no-cpu-baseline --steps 20 --order sum_first $o 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms', round(d['ms_per_step'],4), d['stages_ms'], d['checksum'])"
done
python tools/train_step_c4.py 2>&1 | tail -2
python -m pytest tests -q -m gpu -x > $OUT/gpu_suite.log 2>&1; tail -5 $OUT/gpu_suite.log
