#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/o_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/o_pytest.log
tail -3 gpurun_out/o_pytest.log
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-host-legs > gpurun_out/o_bench.json 2> gpurun_out/o_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/o_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, 'frac', round(d['roofline']['frac'], 4), d['phases_ms_per_step'], d['config']['distinct_keys'])
PY
