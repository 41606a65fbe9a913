#!/bin/bash
# scan kernel: cheaper flagged path + steady-state loop -- parity subset, then the bench line
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu --timeout=600 2>&1 | tail -4
timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-host-legs --no-config5 2>/dev/null | tail -1 > gpurun_out/r3v_bench.json
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3v_bench.json").read())
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"], "phases", d.get("phases_ms_per_step"))
PY
