"""stdin: a bench.py run; prints ms per step, the stage split and the checksum of its JSON line."""
import json
import sys

d = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print(round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d.get("stages_ms", {}).items()}, d.get("checksum"),
      {k: d[k] for k in ("fwd_bwd_ms", "replayed_fwd_bwd_ms", "replay_note") if k in d})
