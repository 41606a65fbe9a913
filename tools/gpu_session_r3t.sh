#!/bin/bash
# radix sort scatter variants: kernel times from rocprofv3 (100 M pairs), then the stage II tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3t
for v in ${VARIANTS:-main}; do
  if [ $v = main ]; then unset MK_LIBRARY; else export MK_LIBRARY=$GRAFT_REPO_ROOT/metakssd_amd/lib_tuning/$v.so; fi
  echo "=== $v"
  rm -rf gpurun_out/r3t/$v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3t/$v -- python3 tools/bench_sort.py --pairs 100000000 --reps 2 2>&1 | grep -E "rep|ok|differs|Error" 
done
unset MK_LIBRARY
timeout 600 python -m pytest tests/test_gpu_mco.py -x -q -m gpu 2>&1 | tail -3
