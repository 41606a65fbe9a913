#!/bin/bash
# t_e2e with the framers started before the HIP runtime is up; CLI parity tests
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_golden.py tests/test_gpu_cli.py -x -q -m gpu --timeout=600 2>&1 | tail -3
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config5 2>/dev/null | tail -1 > gpurun_out/r3e2e_bench.json
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3e2e_bench.json").read())
t = d["t_e2e"]
print({k: t[k] for k in ("gbases_s", "gbases_s_wall", "gbases_s_excl_init", "seconds")})
print(t["timeline_s"])
print(t["all_runs"])
PY
