timeout 600 python bench.py --config c3 --no-cpu-baseline 2>&1 | grep -v "amdgpu.ids" | grep -i "warn\|replay\|error" | head -10
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import sys, torch, warnings
warnings.simplefilter("always")
sys.argv = ["bench.py", "--config", "c3", "--no-cpu-baseline", "--steps", "6"]
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
PY
