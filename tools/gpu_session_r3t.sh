#!/bin/bash
# radix sort variants: kernel times from rocprofv3 (400 M pairs) for the shipped library and lib_tuning/<variant>.so, a correctness run each
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3t
for v in ${VARIANTS:-main}; do
  if [ $v = main ]; then unset MK_LIBRARY; else export MK_LIBRARY=$GRAFT_REPO_ROOT/metakssd_amd/lib_tuning/$v.so; fi
  echo "=== $v"
  timeout 300 python3 tools/bench_sort.py --pairs 30000017 --reps 1 2>&1 | grep -E "ok|differs|Error|assert" | head -3
  rm -rf gpurun_out/r3t/$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3t/$v -- python3 tools/bench_sort.py --pairs 400000000 --reps 2 --no-verify 2>&1 | grep -E "ok|differs|Error"
  f=$(find gpurun_out/r3t/$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r3t_sort_kernel_stats_$v.csv
done
