#!/bin/bash
# HBM traffic of the scan kernel from TCC counters: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots).
# Writes profiles-ready JSON (with the kernel source id bench.py checks) to gpurun_out/scan_traffic.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the JSON carries the id of the SHIPPED kernel source: a run against another library (MK_LIBRARY: make tuning VARIANT=..) would
# file the variant's traffic under that id and bench.py would report it as the shipped kernel's -- refuse
if [ -n "$MK_BENCH_FLAGS$MK_TRAFFIC_KERNEL" ] && [ -z "$MK_TRAFFIC_VARIANT" ]; then echo "tools/pmc_traffic.sh: MK_BENCH_FLAGS / MK_TRAFFIC_KERNEL need MK_TRAFFIC_VARIANT=<tag>" >&2; exit 2; fi
if [ -n "$MK_LIBRARY" ] && [ -z "$MK_TRAFFIC_VARIANT" ]; then echo "tools/pmc_traffic.sh: MK_LIBRARY is set ($MK_LIBRARY): set MK_TRAFFIC_VARIANT=<tag>, the file is then written under the variant's own name" >&2; exit 2; fi
# MK_BENCH_FLAGS: extra bench.py flags (e.g. "--serial-finish"); MK_TRAFFIC_KERNEL: the kernel whose counters are wanted (default
# mk_scan_kernel; e.g. mk_scan_packed_kernel); both only together with MK_TRAFFIC_VARIANT=<tag>, which names the output file
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/traffic_$C -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-host-legs --no-split-leg $MK_BENCH_FLAGS > gpurun_out/traffic_$C.log 2>&1
done
python3 - <<'PY'
import collections, csv, glob, json, os, sys
sys.path.insert(0, '.')
import bench
out = {}
SCAN = os.environ.get("MK_TRAFFIC_KERNEL") or "mk_scan_kernel" 
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/traffic_%s/**/*counter_collection.csv" % C, recursive=True)[0]
    per = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == C:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for kern in (SCAN, "mk_resolve_kernel", "mk_compact_kernel", "mk_layout_kernel", "mk_dump_write_kernel"):
        v = [per[d] for d in per if kern in names[d]]
        out.setdefault(kern, {})[C] = sum(v) / max(1, len(v))
    kname = [names[d] for d in per if SCAN in names[d]]
sc = out[SCAN]
reads_per_launch = bench.CONFIG3_READS  # what `bench.py` without flags scans per launch (one push of the whole workload)
variant = os.environ.get("MK_TRAFFIC_VARIANT")
res = {"kernel": kname[0] if kname else "mk_scan_kernel",
       "kernel_source_id": ("variant[%s] of %s" % (variant, bench.kernel_source_id())) if variant else bench.kernel_source_id(),
       "reads_per_launch": reads_per_launch,
       "FETCH_SIZE_KB": sc["FETCH_SIZE"], "WRITE_SIZE_KB": sc["WRITE_SIZE"],
       "hbm_bytes_per_launch": 2 * sc["FETCH_SIZE"] * 1024 + sc["WRITE_SIZE"] * 1024,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (tools/pmc_traffic.sh); bytes = 2*FETCH_SIZE*1024 + "
                 "WRITE_SIZE*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM "
                 "section; verified in round 1 on a known byte count with this kernel's staging pattern)",
       "other_kernels_KB": {k: v for k, v in out.items() if k != SCAN}}
json.dump(res, open(("gpurun_out/scan_traffic_%s.json" % variant.replace(" ", "_").replace("/", "_")) if variant else "gpurun_out/scan_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
