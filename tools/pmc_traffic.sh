#!/bin/bash
# HBM traffic of the scan kernel from TCC counters: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/traffic_$C -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/traffic_$C.log 2>&1
done
python3 - <<'PY'
import collections, csv, glob, json
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/traffic_%s/runc/*counter_collection.csv" % C)[0]
    per = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == C:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for kern in ("mk_scan_kernel", "mk_resolve_kernel", "mk_compact_kernel"):
        v = [per[d] for d in per if kern in names[d]]
        out.setdefault(kern, {})[C] = sum(v) / max(1, len(v))
print(json.dumps(out))
PY
