#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/y_bench.json 2> gpurun_out/y_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/y_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, 'roofline', d['roofline'], d['phases_ms_per_step'])
print('t_e2e', {k: d['t_e2e'].get(k) for k in ('gbases_s', 'seconds', 'init_s', 'gbases_s_wall', 'wall_s', 'sketch_equals_resident_run')})
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
