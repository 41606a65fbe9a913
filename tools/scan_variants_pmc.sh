#!/bin/bash
# SQ counters (three rocprofv3 --pmc passes, kernel-trace only: tools/pmc_scan.sh) of the variants of tools/scan_variants.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/scan_variants; mkdir -p $O
pmc() { # tag, library dir, extra env
  local tag=$1 lib=$2; shift 2
  export MK_LIBRARY=$PWD/metakssd_amd/lib_tuning/$lib/libmetakssd_hip.so
  for kv in "$@"; do export "$kv"; done
  bash tools/pmc_scan.sh > $O/pmc_$tag.txt 2>&1
  for kv in "$@"; do unset "${kv%%=*}"; done
  unset MK_LIBRARY
}
pmc base base
pmc abl3 abl3
pmc zf8192_2x512 zf8192 MK_SCAN_THREADS=512 MK_SCAN_WGS_PER_CU=2 "MK_BENCH_FLAGS=--cand-cap 32768"
