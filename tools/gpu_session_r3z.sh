#!/bin/bash
# round 3, final session: whole GPU suite, the default bench line, kernel statistics + counters of the same command, fuzz campaign
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ -n "$SKIP_SUITE" ] || timeout 2400 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -15 > gpurun_out/r3z_pytest.log
cat gpurun_out/r3z_pytest.log
timeout 1500 python bench.py > gpurun_out/r3z_bench.json 2> gpurun_out/r3z_bench.err
tail -c 1500 gpurun_out/r3z_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3z_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3z_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-legs > $GRAFT_REPO_ROOT/gpurun_out/r3z_prof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3z_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r3z_kernel_stats.csv && head -16 $f | cut -c1-200
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh > gpurun_out/r3z_traffic.log 2>&1; tail -2 gpurun_out/r3z_traffic.log | cut -c1-600
bash tools/pmc_scan.sh > gpurun_out/r3z_pmc_scan.log 2>&1
python3 tools/pmc_summary.py > gpurun_out/r3z_scan_pmc.txt 2>&1; cat gpurun_out/r3z_scan_pmc.txt
bash tools/gpu_session_fz.sh > gpurun_out/r3z_fz.log 2>&1; tail -12 gpurun_out/r3z_fz.log | cut -c1-300
# stage II's sort alone at the size of one L3K10 component build (400 M pairs): kernel times
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3z_sort
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3z_sort -- python3 $GRAFT_REPO_ROOT/tools/bench_sort.py --pairs 400000000 --reps 2 --no-verify > $GRAFT_REPO_ROOT/gpurun_out/r3z_sort.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3z_sort -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r3z_sort_kernel_stats.csv && grep mk_rs_ $f | cut -c1-160
cd $GRAFT_REPO_ROOT
