#!/bin/bash
# A/B on one box: the library in the tree against metakssd_amd/lib_tuning/libmetakssd_hip_prev.so, three rounds
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in "cur" "prev"; do
lib=""
[ $v = prev ] && lib=$GRAFT_REPO_ROOT/metakssd_amd/lib_tuning/libmetakssd_hip_prev.so
MK_LIBRARY=$lib timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] scan_ms', round(d['roofline']['avg_launch_ms'],3), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, 'ms', round(d['ms_per_step'],3), 'distinct', d['config']['distinct_keys'])"
done
done
