#!/bin/bash
cd $GRAFT_REPO_ROOT
make -s -C metakssd_amd/csrc tuning TUNING_OUT=/tmp/mk_tuning VARIANT="-DMK_TUNING_BUILD=1" || exit 1
for g in 128 256 384 512; do
MK_RESOLVE_GRID=$g MK_LIBRARY=/tmp/mk_tuning/libmetakssd_hip.so python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resolve grid $g', {k: round(v,3) for k,v in d['phases_ms_per_step'].items()})"
done
