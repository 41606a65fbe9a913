#!/bin/bash
# A round's profiler evidence in one gpurun call.  Every file name carries the kernel source id (bench.kernel_source_id(): sha256 over
# the kernel sources), the same id bench.py prints in `roofline.kernel_source_id`: gpurun_out/<round>_<id>_*  ->  copy to profiles/.
#   <R>_<id>_kernel_stats.csv                 rocprofv3 --kernel-trace --stats of the HEADLINE flow (one engine, one queue), passes pipelined
#   <R>_<id>_kernel_stats_serial_finish.csv   the same with --serial-finish (nothing beside the scan kernel)
#   <R>_<id>_kernel_stats_split.csv           the split-queue side flow alone (--split-leg-only)
#   <R>_<id>_bench_*.json                     the bench lines of those three runs
#   <R>_<id>_pmc_one_queue.txt / _pmc_split.txt   SQ counters of mk_scan_kernel and mk_resolve_kernel (256 CUs / 224 + 32 CUs), separate --pmc passes
#   <R>_<id>_split_kernel_trace.csv           kernel trace of 20 split-queue passes
#   <R>_<id>_scan_traffic.json                TCC FETCH_SIZE / WRITE_SIZE of the scan kernel (tools/pmc_traffic.sh)
#   <R>_<id>_config5_<geometry>_kernel_stats.csv   tools/profile_config5.sh
# usage: bash tools/profile_round.sh [round tag, default r06]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r06}
ID=$(python3 -c "import bench; print(bench.kernel_source_id())")
O=gpurun_out/${R}_profile; mkdir -p $O
P=$O/${R}_${ID}
echo "kernel source id $ID -> $P_*"
run_stats() { # tag, extra bench flags
  local tag=$1; shift
  rm -rf gpurun_out/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 30 --warmup 3 --no-host-legs --no-cpu-baseline "$@" > ${P}_bench_$tag.json 2> $O/$tag.err
  local f=$(ls -t $(find gpurun_out/prof_$tag -name "*kernel_stats.csv") | head -1)
  cp "$f" ${P}_$tag.csv; echo "== $tag"; head -8 ${P}_$tag.csv | cut -c1-160
}
run_stats kernel_stats --no-split-leg
run_stats kernel_stats_serial_finish --no-split-leg --serial-finish
run_stats kernel_stats_split --split-leg-only
# kernel trace of 20 split-queue passes (timeline: scan on 224 units, resolve + compaction + clear beside the next scan)
rm -rf gpurun_out/prof_split_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_split_trace -- python3 bench.py --steps 20 --warmup 5 --no-host-legs --no-cpu-baseline --split-leg-only > $O/split_trace.json 2> $O/split_trace.err
cp "$(ls -t $(find gpurun_out/prof_split_trace -name "*kernel_trace.csv") | head -1)" ${P}_split_kernel_trace.csv
# counters: three passes each, kernel-trace only
bash tools/pmc_scan.sh > ${P}_pmc_one_queue.txt 2>&1
MK_BENCH_FLAGS="--split-leg-only" bash tools/pmc_scan.sh > ${P}_pmc_split.txt 2>&1
bash tools/pmc_traffic.sh > $O/scan_traffic.log 2>&1; cp gpurun_out/scan_traffic.json ${P}_scan_traffic.json; head -8 ${P}_scan_traffic.json
bash tools/profile_config5.sh > $O/config5_profile.log 2>&1
cp gpurun_out/c5_L3K10_kernel_stats.csv ${P}_config5_L3K10_kernel_stats.csv; cp gpurun_out/c5_L2K11_kernel_stats.csv ${P}_config5_L2K11_kernel_stats.csv
ls -la $O
