#!/bin/bash
# The clock the scan kernel runs at: GRBM_GUI_ACTIVE (cycles the XCDs were busy, summed over the 8) / its duration, on the whole device and
# on a part of it (experiment build: MK_TUNE_CUS).  make -C metakssd_amd/csrc tuning TUNING_OUT=../lib_tuning/base
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MK_LIBRARY=metakssd_amd/lib_tuning/base/libmetakssd_hip.so
: > gpurun_out/scan_clock.txt
for cus in 256 128 64 32; do
  rm -rf gpurun_out/clock_$cus
  if [ $cus = 256 ]; then unset MK_TUNE_CUS; else export MK_TUNE_CUS=$cus; fi
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/clock_$cus -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --no-traffic --serial-finish > gpurun_out/clock_$cus.log 2>&1
  python3 - $cus <<'PY' | tee -a gpurun_out/scan_clock.txt
import csv, glob, sys, collections
cus = sys.argv[1]
d = "gpurun_out/clock_%s" % cus
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "mk_scan_kernel" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = acc[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in list(acc.items())[-2:]:
    ns = dur.get(k)
    if not ns: continue
    g = v.get("GRBM_GUI_ACTIVE", 0.0)
    wc, w = v.get("SQ_WAVE_CYCLES", 0.0), v.get("SQ_WAVES", 0.0)
    print("%s CUs: scan %.3f ms  GRBM_GUI_ACTIVE %.0f = %.3f GHz if summed over 8 XCDs  SQ_WAVE_CYCLES %.0f / %d waves: x4 / duration = %.3f GHz  SQ_BUSY_CYCLES %.0f" % (
        cus, ns / 1e6, g, g / 8 / ns, wc, w, (wc * 4 / w / ns) if w else 0, v.get("SQ_BUSY_CYCLES", 0.0)))
PY
done
