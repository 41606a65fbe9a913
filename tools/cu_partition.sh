#!/bin/bash
# What the scan and the resolve kernel take on a PART of the chip (would the resolve of one pass fit beside the scan of the next
# on compute units of its own?).  Needs the experiment build: make -C metakssd_amd/csrc tuning TUNING_OUT=../lib_tuning/base
# The engine sizes every grid for MK_TUNE_CUS units (and its own queue gets a CU mask from bit MK_TUNE_CU_FIRST on -- bench.py runs the engine on
# torch's stream, so here it is the grids that confine the kernels; the other units are idle).
cd $GRAFT_REPO_ROOT
export MK_LIBRARY=metakssd_amd/lib_tuning/base/libmetakssd_hip.so
out=gpurun_out/cu_partition.txt; : > $out
run() { # label, env...
  label=$1; shift
  line=$(env "$@" python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-host-legs --no-traffic --serial-finish 2>gpurun_out/cu_partition.err | tail -1)
  echo "$label $(echo "$line" | python3 -c 'import json,sys; j=json.loads(sys.stdin.read()); p=j["phases_ms_per_step"]; print("ms_per_step %.3f scan %.3f resolve %.3f finish %.3f clear %.3f" % (j["ms_per_step"], p["scan"], p["resolve"], p["finish"], p["clear"]))' 2>&1)" | tee -a $out
  grep -h "tuning\]" gpurun_out/cu_partition.err | head -1 >> $out
}
run "all 256 CUs      " MK_X=1
run "240 CUs (0..239) " MK_TUNE_CUS=240
run "224 CUs (0..223) " MK_TUNE_CUS=224
run "32 CUs (0..31)   " MK_TUNE_CUS=32
run "32 CUs (224..255)" MK_TUNE_CUS=32 MK_TUNE_CU_FIRST=224
run "16 CUs (0..15)   " MK_TUNE_CUS=16
run "16 CUs (240..255)" MK_TUNE_CUS=16 MK_TUNE_CU_FIRST=240
run "8 CUs (0..7)     " MK_TUNE_CUS=8
