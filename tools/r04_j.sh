OUT=gpurun_out/r04j
mkdir -p $OUT
python tools/debug_refacc.py > $OUT/debug_refacc.log 2>&1
python -m pytest tests/test_gpu_harness.py -x -q -m gpu > $OUT/test_harness.log 2>&1; echo "harness tests rc=$?" >> $OUT/rc.log
for fork in start fmlp; do for B in 256 512; do
  echo "== fork $fork buckets $B" >> $OUT/emu_variants.txt
  python bench.py --traffic committed --sustain-seconds 0 --steps 50 --warmup 5 --no-cpu-baseline --emulate-world 8 --partition halo --share-fork $fork --index-buckets $B 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wall', round(d['ms_per_step'],4), 'device', d['step_ms_device'], 'graphs' if d['share_replayed_from_hipgraphs'] else 'eager', d['share_graph_note'] or '', d['share_guard_tripped'], 'checksum', d['checksum'])" >> $OUT/emu_variants.txt
done; done
python bench.py --traffic committed --sustain-seconds 0 --steps 50 --warmup 5 --no-cpu-baseline --emulate-world 8 --partition halo --share-graph off --index-buckets 256 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('eager B256 wall', round(d['ms_per_step'],4), 'device', d['step_ms_device'], d['stages_ms'])" >> $OUT/emu_variants.txt
python -m pytest tests/test_gpu_multirank.py -q -m gpu > $OUT/test_multirank.log 2>&1; echo "multirank tests rc=$?" >> $OUT/rc.log
cat $OUT/rc.log; cat $OUT/debug_refacc.log | tail -12; tail -5 $OUT/test_harness.log; cat $OUT/emu_variants.txt; tail -8 $OUT/test_multirank.log | cut -c1-300
