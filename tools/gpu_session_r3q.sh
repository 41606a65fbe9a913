#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3q_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3q_prof -- python3 $GRAFT_REPO_ROOT/tools/bench_search.py --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/r3q_prof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3q_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r3q_search_kernel_stats.csv && head -14 $f | cut -c1-160
grep '"op"' $GRAFT_REPO_ROOT/gpurun_out/r3q_prof.log | cut -c1-300
