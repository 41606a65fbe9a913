#!/bin/bash
# What the HBM stream costs the scan kernel (round 5): experiment builds against the shipped kernel, one queue, one box, two rounds.
#   base    shipped            nt      the rows through non-temporal loads (make tuning VARIANT=-DMK_SCAN_NT=1)
#   ablmem  every wave stages its FIRST tile again and again (cache-resident; results wrong): the kernel without its HBM stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in ${VARIANTS:-base nt ablmem}; do
    MK_LIBRARY=$PWD/metakssd_amd/lib_tuning/$v/libmetakssd_hip.so python3 bench.py --steps 100 --split-cus 0 --no-host-legs --no-cpu-baseline --no-traffic 2>/dev/null |
      python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
p=d['phases_ms_per_step']
print('[$v] scan_ms %.4f resolve_ms %.4f ms/step %.3f distinct %s' % (d['roofline']['avg_launch_ms'], p['resolve'], d['ms_per_step'], d['config']['distinct_keys']))"
  done
done
