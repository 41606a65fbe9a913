#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -q --timeout=300 tests/test_gpu_parity.py -k "two_halves or sparse or front" 2>&1 | tail -8
timeout 900 python -m pytest -m gpu -q --timeout=600 tests/test_dist_gloo.py tests/test_gpu_fullsize.py 2>&1 | tail -8
for i in 1 2; do
timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, 'scan', round(d['roofline']['avg_launch_ms'],3), 'distinct', d['config']['distinct_keys'])"
done
