#!/bin/bash
# the scan kernel's duration without side-stream work beside it: live (HIP events) and under rocprofv3, same command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-legs --serial-finish 2>/dev/null | tail -1 > gpurun_out/r3y_serial_live.json
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3y_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3y_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-legs --serial-finish > $GRAFT_REPO_ROOT/gpurun_out/r3y_prof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3y_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r3y_kernel_stats_serial.csv && head -8 $f | cut -c1-160
grep -v "^W2026\|^E2026" $GRAFT_REPO_ROOT/gpurun_out/r3y_prof.log | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/r3y_serial_under_rocprof.json
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json
for f in ("gpurun_out/r3y_serial_live.json", "gpurun_out/r3y_serial_under_rocprof.json"):
    d = json.loads(open(f).read())
    print(f, "scan avg_launch_ms", d["roofline"]["avg_launch_ms"], "ms/step", d["ms_per_step"])
PY
