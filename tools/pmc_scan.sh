#!/bin/bash
# PMC passes over the bench (each pass its own run, kernel-trace only) -> gpurun_out/pmc_*/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT

P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD"
P3="GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_$i   # (a directory left by an earlier call would hand its counters to this one)
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-host-legs ${MK_BENCH_FLAGS:---no-split-leg} > gpurun_out/pmc_$i.log 2>&1
  f=$(find gpurun_out/pmc_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "mk_scan_kernel" in k or "mk_resolve_kernel" in k:
        for c, v in acc[k].items():
            print(k, c, v / max(1, n[(k, c)]))
PY
done
