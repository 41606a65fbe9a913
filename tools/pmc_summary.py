#!/usr/bin/env python3
"""summarise gpurun_out/pmc_*/ counter CSVs for one kernel (per launch and per unit of work)
usage: pmc_summary.py [units per launch = 781250 tiles of 64 reads] [kernel name substring = scan_kernel]"""
import collections, csv, glob, sys
tiles = float(sys.argv[1]) if len(sys.argv) > 1 else 781250.0
kname = sys.argv[2] if len(sys.argv) > 2 else 'scan_kernel'
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('gpurun_out/pmc_*/runc/*counter_collection.csv'):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kname in r['Kernel_Name']:
            per[(r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (d, c), v in per.items():
        acc[c] += v; n[c] += 1
for k in sorted(acc):
    v = acc[k] / n[k]
    print(f"{k:28s} {v:16.0f}  per tile {v / tiles:10.1f}")
