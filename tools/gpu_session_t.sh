#!/bin/bash
# side-stream table fills: whole GPU suite + bench phases
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_pytest.log
tail -5 gpurun_out/t_pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-host-legs 2>gpurun_out/t_bench.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scan_ms', round(d['roofline']['avg_launch_ms'],3), d['phases_ms_per_step'], 'Gb/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'distinct', d['config']['distinct_keys'])"
done
tail -3 gpurun_out/t_bench.err
