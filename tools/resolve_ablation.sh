#!/bin/bash
# Round 5: where mk_resolve_kernel's 0.21 ms go -- ablation builds (make tuning VARIANT=-DMK_ABL_R=n, results WRONG by construction, timing only):
#   base   shipped            rabl1  no .shuf look-up, no table update (ring 2 drained into nothing)
#   rabl2  + no canonical k-mer / accept-bit look-up (ring 1 drained without a global access)       rabl3  + no filter test: records fetched, nothing else
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in base rabl1 rabl2 rabl3; do
    MK_LIBRARY=$PWD/metakssd_amd/lib_tuning/$v/libmetakssd_hip.so python3 bench.py --steps 100 --no-host-legs --no-cpu-baseline 2>/dev/null |
      python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
p=d['phases_ms_per_step']
print('[$v] resolve_ms %.4f scan_ms %.3f ms/step %.3f distinct %s' % (p['resolve'], d['roofline']['avg_launch_ms'], d['ms_per_step'], d['config']['distinct_keys']))"
  done
done
