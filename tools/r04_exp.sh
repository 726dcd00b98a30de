#!/bin/bash
OUT=gpurun_out/r04exp; mkdir -p $OUT
python tools/lookup_ab.py > $OUT/ab.jsonl 2> $OUT/err.log
python - <<'PY'
import json
for l in open('gpurun_out/r04exp/ab.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('variant','mode','ms','ms_min','TBps')})
PY
