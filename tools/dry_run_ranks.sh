# Dry run of the multi-rank bench on a 1-GPU box: P processes share cuda:0, collectives over gloo.  Checks that the
# partitions agree with the single-rank checksum (times are meaningless: the ranks share one device).
P=${1:-2}
python bench.py --traffic committed --sustain-seconds 0 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('single', 1, d['checksum'])"
for part in ${PARTS:-halo vertex feature exchange}; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $P --steps 2 --warmup 1 \
    --backend gloo --same-device --no-cpu-baseline --partition $part 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$part', d['n_gpus'], d['checksum'], d['config']['partition'], d['config']['exchange'])"
done
