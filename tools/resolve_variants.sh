#!/bin/bash
# Round 5: mk_resolve_kernel with 2 / 4 (shipped) / 6 / 8 blocks of 64 records a wave and round (make tuning VARIANT=-DMK_RESOLVE_UNROLL=<n>u
# -> metakssd_amd/lib_tuning/ru<n>/): resolve ms per step from the engine's events, HBM-resident config 3, two rounds on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in base ru2 ru6 ru8; do
    MK_LIBRARY=$PWD/metakssd_amd/lib_tuning/$v/libmetakssd_hip.so python3 bench.py --steps 100 --no-host-legs --no-cpu-baseline 2>/dev/null |
      python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
p=d['phases_ms_per_step']
print('[$v] resolve_ms %.4f scan_ms %.3f ms/step %.3f distinct %s' % (p['resolve'], d['roofline']['avg_launch_ms'], d['ms_per_step'], d['config']['distinct_keys']))"
  done
done
