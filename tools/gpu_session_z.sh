#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/z_pytest.log 2>&1; tail -3 gpurun_out/z_pytest.log
python tools/fuzz_parity.py --cases 500 --seed 3001 --big 2>&1 | tail -1
python tools/fuzz_parity.py --cases 800 --seed 3002 --sparse 1 2>&1 | tail -1
GENOMES=1024 THREADS=64 REF_GENOMES=1 python tools/bench_config5.py 2>&1 | grep -v "^{" | grep -v reference | head -8
python bench.py --steps 100 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()})"
