#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest -m gpu -q --timeout=600 tests/test_gpu_parity.py tests/test_gpu_fuzz.py 2>&1 | tail -6
timeout 900 python -m pytest -m gpu -q --timeout=600 tests/test_golden.py -k "many_small or fasta or pool or several" 2>&1 | tail -4
timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, 'distinct', d['config']['distinct_keys'])"
python tools/bench_config5_variants.py 2>&1 | grep tool | cut -c1-330
