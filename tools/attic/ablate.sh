#!/bin/bash
# experimental: engine variants built with extra -D flags into a SCRATCH library (make tuning; the shipped library is not
# touched) and the bench run against each through MK_LIBRARY
# usage: VARIANTS="none|-DMK_SOMETHING=1" bash tools/ablate.sh
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VS <<< "${VARIANTS:-none}"
i=0
for v in "${VS[@]}"; do
  i=$((i+1))
  out=/tmp/mk_variant_$i
  if [ "$v" = "none" ]; then lib=""; else
    make -s -C metakssd_amd/csrc tuning TUNING_OUT=$out VARIANT="$v" || { echo "variant [$v] does not build"; continue; }
    lib=$out/libmetakssd_hip.so
  fi
  MK_LIBRARY=$lib python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-host-legs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v] scan_ms', round(d['roofline']['avg_launch_ms'],3), 'resolve', round(d['phases_ms_per_step']['resolve'],3), 'finish', round(d['phases_ms_per_step']['finish'],3), 'Gb/s', round(d['value'],1), 'distinct', d['config']['distinct_keys'])"
done
