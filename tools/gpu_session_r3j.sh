#!/bin/bash
# round 3, session j: SQ counters of the scan kernel for the shipped library and two ablation builds (results WRONG by construction:
# abl2 = no mask-table reads, abl5 = all probes at conflict-free addresses); ratios per variant into one text file
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r3j_scan_variant_counters.txt
: > $OUT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD"
for V in cur abl2 abl5; do
  lib=""
  [ $V != cur ] && lib=$GRAFT_REPO_ROOT/metakssd_amd/lib_tuning/$V.so
  ms=$(MK_LIBRARY=$lib python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],3))")
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1))
    rm -rf gpurun_out/pmcv_${V}_$i
    MK_LIBRARY=$lib rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmcv_${V}_$i -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-host-legs > gpurun_out/pmcv_${V}_$i.log 2>&1
  done
  python3 - $V $ms >> $OUT <<'PY'
import csv, glob, sys, collections
V, ms = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('gpurun_out/pmcv_%s_*/**/*counter_collection.csv' % V, recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'scan_kernel' in r['Kernel_Name']:
            per[(r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (d, c), v in per.items():
        acc[c] += v; n[c] += 1
a = {k: acc[k] / n[k] for k in acc}
g = lambda k: a.get(k, float('nan'))
print("[%s] scan %s ms (HIP events, unprofiled run)  LDS conflict/active %.3f  VALU active/wave-cycles x4 %.3f  LDS insts/VALU insts %.3f  "
      "wait_any/wave-cycles %.3f  wait_inst_lds/wave-cycles %.3f  VALU insts per tile(seen waves) %.1f  LDS insts per tile %.1f" % (
      V, ms, g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'), 4 * g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'),
      g('SQ_INSTS_LDS') / g('SQ_INSTS_VALU'), g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'), g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES'),
      g('SQ_INSTS_VALU') / 781250.0, g('SQ_INSTS_LDS') / 781250.0))
PY
done
cat $OUT
