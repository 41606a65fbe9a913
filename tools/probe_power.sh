#!/bin/bash
# what an ordinary user can read about the GPU's clock and power on the box (sysfs / rocm-smi), idle and under the bench's scan
cd $GRAFT_REPO_ROOT
for d in /sys/class/drm/card*/device; do
  [ -e $d/pp_dpm_sclk ] || continue
  echo "== $d"; cat $d/pp_dpm_sclk 2>&1 | head -5
  for h in $d/hwmon/hwmon*; do
    for f in power1_cap power1_cap_max power1_cap_default power1_average power1_input freq1_input freq2_input temp1_input; do
      [ -e $h/$f ] && echo "$f $(cat $h/$f 2>&1)"
    done
  done
done
rocm-smi --showpower --showmaxpower --showclocks 2>&1 | head -30
python3 bench.py --steps 12000 --no-host-legs --no-cpu-baseline --no-traffic --no-one-queue --split-cus 0 > /dev/null 2>&1 &
BP=$!
sleep 12   # torch import + setup; the 12000 passes take 28 s
for i in $(seq 1 40); do
  for d in /sys/class/drm/card*/device; do
    [ -e $d/pp_dpm_sclk ] || continue
    for h in $d/hwmon/hwmon*; do echo "t=$i $(basename $(dirname $d)) under load: freq1 $(cat $h/freq1_input 2>/dev/null) power_avg $(cat $h/power1_average 2>/dev/null) power_in $(cat $h/power1_input 2>/dev/null) temp $(cat $h/temp1_input 2>/dev/null)"; done
  done
  sleep 1
done
wait $BP
