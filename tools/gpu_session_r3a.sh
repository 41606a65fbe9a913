#!/bin/bash
# round 3, session a: where does the scan kernel's time go?  Ablation libraries (make tuning VARIANT=-DMK_ABL=n, wrong results,
# timing only) against the shipped one on one box, two rounds.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r3a_ablation.txt
: > $out
for i in 1 2; do
for v in cur abl1 abl2 abl3 abl4 abl5; do
lib=""
[ $v != cur ] && lib=$GRAFT_REPO_ROOT/metakssd_amd/lib_tuning/$v.so
MK_LIBRARY=$lib timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] scan_ms', round(d['roofline']['avg_launch_ms'],3), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, 'ms', round(d['ms_per_step'],3), 'distinct', d['config']['distinct_keys'])" | tee -a $out
done
done
