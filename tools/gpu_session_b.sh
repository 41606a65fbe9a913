#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/b_pytest.log
tail -4 gpurun_out/b_pytest.log
./tools/probe_h2d > gpurun_out/b_probe_h2d.json 2>&1; cat gpurun_out/b_probe_h2d.json
timeout 1200 python tools/bench_e2e.py --reads 50000000 --threads 12,16,24,32 --chunks 4,8 --reps 2 --variants ";--keep-pages" --out gpurun_out/b_e2e_sweep.json > gpurun_out/b_e2e_sweep.log 2>&1
grep SUMMARY gpurun_out/b_e2e_sweep.log
