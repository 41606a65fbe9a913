#!/bin/bash
# A/B on one box: current library (front table on / off) against the previous commit's library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
for v in "cur" "cur0" "prev"; do
fb=""; lib=""
[ $v = cur0 ] && fb="--front-bits 0"
[ $v = prev ] && lib=$GRAFT_REPO_ROOT/metakssd_amd/lib_tuning/libmetakssd_hip_prev.so
MK_LIBRARY=$lib timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-host-legs $fb 2>gpurun_out/u_bench.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] scan_ms', round(d['roofline']['avg_launch_ms'],3), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, 'Gb/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'distinct', d['config']['distinct_keys'])"
done
done
tail -3 gpurun_out/u_bench.err
