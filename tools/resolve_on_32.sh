#!/bin/bash
# mk_resolve_kernel on 32 compute units (the second queue of MK_OPT_SPLIT_CUS): blocks a wave and round, and the ablation stages of
# tools/resolve_ablation.sh, with the engine's one queue under a 32-unit mask (make tuning builds, MK_TUNE_CUS=32)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MK_TUNE_CUS=32
for round in 1 2; do
  for v in base ru2 ru8 rabl1 rabl2 rabl3; do
    MK_LIBRARY=$PWD/metakssd_amd/lib_tuning/$v/libmetakssd_hip.so python3 bench.py --steps 10 --warmup 2 --serial-finish --no-host-legs --no-cpu-baseline --no-traffic 2>/dev/null |
      python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
p=d['phases_ms_per_step']
print('[$v] 32 units: resolve_ms %.4f scan_ms %.3f distinct %s' % (p['resolve'], d['roofline']['avg_launch_ms'], d['config']['distinct_keys']))"
  done
done
