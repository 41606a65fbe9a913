#!/bin/bash
# ablation 6 (no call in the scan kernel's overflow path) against the shipped library, three rounds on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
for v in cur abl6; do
lib=""
[ $v != cur ] && lib=$GRAFT_REPO_ROOT/metakssd_amd/lib_tuning/$v.so
MK_LIBRARY=$lib timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] scan_ms', round(d['roofline']['avg_launch_ms'],3), 'ms/step', round(d['ms_per_step'],3), 'distinct', d['config']['distinct_keys'])" | tee -a gpurun_out/r3o_abl6.txt
done
done
