#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden.py -m gpu -x -q > gpurun_out/z_pytest.log 2>&1; tail -3 gpurun_out/z_pytest.log
python tools/fuzz_parity.py --cases 800 --seed 4242 2>&1 | tail -1
python tools/probe_ragged_reads.py 2>&1 | tail -1
for i in 1 2; do
for v in "cur" "prev"; do
lib=""
[ $v = prev ] && lib=$GRAFT_REPO_ROOT/metakssd_amd/lib_tuning/libmetakssd_hip_prev.so
MK_LIBRARY=$lib timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-host-legs 2>gpurun_out/u_bench.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] scan_ms', round(d['roofline']['avg_launch_ms'],3), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, 'Gb/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'distinct', d['config']['distinct_keys'])"
done
done
