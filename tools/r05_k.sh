timeout 600 python tools/debug_replay.py 2>&1 | grep -v "Warning\|warn" | tail -12
