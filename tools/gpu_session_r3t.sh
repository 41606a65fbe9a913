#!/bin/bash
# radix sort: kernel times from rocprofv3 (400 M pairs), then the stage II tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3t
rm -rf gpurun_out/r3t/main
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3t/main -- python3 tools/bench_sort.py --pairs 400000000 --reps 2 --no-verify 2>&1 | grep -E "rep|ok|differs|Error"
f=$(find gpurun_out/r3t/main -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r3t_sort_kernel_stats.csv && grep -E "mk_rs_" "$f" </dev/null | cut -d, -f1-6 | cut -c1-40,150-
timeout 600 python -m pytest tests/test_gpu_mco.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/fuzz_search.py --seconds 60 --seed 5001 2>&1 | tail -1
