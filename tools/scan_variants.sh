#!/bin/bash
# Round 5: the two scan-kernel variants VERDICT r04 item 3 names, against the shipped kernel ON ONE BOX (make tuning builds in
# metakssd_amd/lib_tuning/<variant>/, see metakssd_amd/csrc/Makefile `tuning`):
#   base    the shipped kernels, built with -DMK_TUNING (the experiment knobs compiled in, nothing changed)
#   abl3    3(i)  mask-table reads at conflict-free LDS addresses (bank := lane) for one more VALU a probe -- the timing of a mask
#                 table that costs no bank conflicts (a per-lane replica, ds_bpermute of a register-held table); results WRONG by design
#   zf8192  3(ii) a 32 KiB pair filter (the word index drops the key's top bit), run as shipped (1 x 1024 threads a CU) and as TWO
#                 512-thread workgroups a CU (MK_SCAN_THREADS=512 MK_SCAN_WGS_PER_CU=2); results RIGHT (the resolve kernel is exact)
# per variant: bench.py (HBM-resident config 3, 100 steps: scan / resolve ms from the engine's events, pipelined and --serial-finish), the
# candidate records a launch makes, and the SQ counters of tools/pmc_scan.sh.  Output: gpurun_out/scan_variants/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/scan_variants; mkdir -p $O
run() { # tag, library dir, extra env (XFLAGS=<extra bench.py flags>)
  local tag=$1 lib=$2; shift 2
  local xf=""
  for kv in "$@"; do case $kv in XFLAGS=*) xf=${kv#XFLAGS=};; esac; done
  for mode in "" "--serial-finish"; do
    env "$@" MK_LIBRARY=$PWD/metakssd_amd/lib_tuning/$lib/libmetakssd_hip.so python3 bench.py --steps 100 --no-host-legs --no-cpu-baseline $xf $mode 2> $O/${tag}${mode:+_serial}.err |
      python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
p=d['phases_ms_per_step']
print('[$tag${mode:+ serial}] scan_ms %.3f resolve_ms %.3f ms/step %.3f Gbases/s %.0f distinct %s roofline.frac %.3f' % (d['roofline']['avg_launch_ms'], p['resolve'], d['ms_per_step'], d['value'], d['config']['distinct_keys'], d['roofline']['frac']))"
  done
  env "$@" MK_COUNT_RECORDS=1 MK_LIBRARY=$PWD/metakssd_amd/lib_tuning/$lib/libmetakssd_hip.so python3 bench.py --steps 1 --warmup 0 --no-host-legs --no-cpu-baseline $xf 2>&1 >/dev/null | grep "scan records" | head -1 | sed "s/^/[$tag] /"
}
for round in 1 2; do
  run base base
  run abl3 abl3
  # (the 32 KiB filter flags three times the windows: 32768 records a wave instead of 8192, so that no wave's buffer fills up and
  #  falls back to resolving inline -- the first run of this script measured exactly that, 7.7 ms a launch)
  run zf8192_1x1024 zf8192 "XFLAGS=--cand-cap 32768"
  run zf8192_2x512 zf8192 MK_SCAN_THREADS=512 MK_SCAN_WGS_PER_CU=2 "XFLAGS=--cand-cap 32768"
  run base_1x512_64k base MK_SCAN_THREADS=512
done
