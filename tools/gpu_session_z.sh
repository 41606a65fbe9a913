#!/bin/bash
# 150-base reads: one-pass staging on / off, column block widths
cd $GRAFT_REPO_ROOT
make -s -C metakssd_amd/csrc tuning TUNING_OUT=/tmp/mk_tuning VARIANT="-DMK_TUNING_BUILD=1" || exit 1
for v in "MK_SCAN_ONEPASS=1" "MK_SCAN_ONEPASS=0" "MK_SCAN_CB=64" "MK_SCAN_CB=48" "MK_SCAN_CB=32"; do
env $v MK_DEBUG=1 MK_LIBRARY=/tmp/mk_tuning/libmetakssd_hip.so python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-host-legs 2>/tmp/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, d['config']['distinct_keys'])"
grep "scan cfg" /tmp/err.log | head -1
done
