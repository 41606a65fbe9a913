#!/bin/bash
# round-2 GPU session A: parity suite, start-up probe, t_e2e sweep, the bench line with the host-inclusive legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc > gpurun_out/a_box.txt; free -g >> gpurun_out/a_box.txt; df -h /dev/shm >> gpurun_out/a_box.txt; lscpu | head -25 >> gpurun_out/a_box.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/a_pytest.log
tail -5 gpurun_out/a_pytest.log
for i in 1 2 3; do ./tools/probe_init; done > gpurun_out/a_probe_init.jsonl 2>&1
timeout 1200 python tools/bench_e2e.py --reads 50000000 --threads 8,16,32,48,64 --chunks 4,8,32 --reps 2 --out gpurun_out/a_e2e_sweep.json > gpurun_out/a_e2e_sweep.log 2>&1
tail -3 gpurun_out/a_e2e_sweep.log
timeout 900 python bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench rc=$?"
cat gpurun_out/a_bench.json | head -c 6000
