#!/bin/bash
# kernel timeline of bench.py's split-queue flow (a few passes): who runs when, from rocprofv3's kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-24}
rm -rf gpurun_out/split_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/split_trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-legs --no-traffic --no-one-queue --split-cus $R > gpurun_out/split_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/split_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 60 kernels before the end of the timed region: find the last mk_scan_kernel and go back
idx = [i for i, r in enumerate(rows) if "mk_scan_kernel" in r["Kernel_Name"]]
lo = idx[-7] if len(idx) >= 7 else 0
hi = idx[-2] + 12 if len(idx) >= 2 else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
out = open("gpurun_out/split_trace.txt", "w")
for r in rows[lo:hi]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    line = "%-40s q%-3s start %9.3f end %9.3f  dur %8.3f ms  grid %s" % (name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6,
                                                       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?")))
    print(line); out.write(line + "\n")
PY
