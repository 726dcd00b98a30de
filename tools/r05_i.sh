# round 5, ninth GPU call: hop-code tiers of the graph slots; timeline of a replayed slot step
mkdir -p gpurun_out/r05i
timeout 900 python -m pytest tests/test_gpu_graphed.py -q -m gpu -x -k "slot" 2>&1 | tail -3
timeout 900 python tools/muta_epoch.py > gpurun_out/r05i/muta_epoch.json 2>/dev/null; cat gpurun_out/r05i/muta_epoch.json | cut -c1-1200
bash tools/graphed_timeline.sh gpurun_out/r05i/tl_muta muta > gpurun_out/r05i/timeline_muta.txt 2>&1; tail -25 gpurun_out/r05i/timeline_muta.txt | cut -c1-170
