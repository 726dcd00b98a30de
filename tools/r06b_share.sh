OUT=gpurun_out/r06b; mkdir -p $OUT
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_rccl.py -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for cut in rows cost; do for fork in start fmlp; do
 echo "== cut $cut fork $fork"
 RANK=2 python bench.py --traffic off --sustain-seconds 0 --steps 200 --warmup 10 --no-cpu-baseline --emulate-world 8 --partition halo --force-dist --cut $cut --share-fork $fork 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wall_ms', round(d['ms_per_step'],4), 'device', d['step_ms_device'], 'rows', d['operand_rows_rank0'], 'pairs', d['config']['stored_pairs_rank0'], 'graphs' if d['share_replayed_from_hipgraphs'] else 'eager', d['share_graph_note'] or '', d.get('owned_rows_rank0'), 'checksum', d['checksum'])"
done; done
RANK=2 bash tools/step_timeline.sh $OUT/share_tl --emulate-world 8 --partition halo --force-dist --share-fork fmlp > $OUT/share_timeline_fmlp.txt 2>&1
cat $OUT/share_timeline_fmlp.txt
