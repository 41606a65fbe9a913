#!/bin/bash
# The round's evidence in one call (run through gpurun; everything lands in gpurun_out/round/, to be copied to profiles/rNN_*):
#   kernel_stats.csv / kernel_stats_serial_finish.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 30 --warmup 3 --no-host-legs
#                                                       --no-cpu-baseline` [--serial-finish] + the bench lines of those runs
#   scan_pmc.txt                                        SQ counters of the scan / resolve kernels (tools/pmc_scan.sh)
#   scan_traffic.json                                   FETCH_SIZE / WRITE_SIZE of the scan kernel (tools/pmc_traffic.sh), keyed to the kernel source id
#   config5_<geometry>_kernel_stats.csv                 tools/profile_config5.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/round; mkdir -p $O
for mode in "" "--serial-finish"; do
  tag=kernel_stats${mode:+_serial_finish}
  rm -rf gpurun_out/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 30 --warmup 3 --no-host-legs --no-cpu-baseline --no-one-queue $mode > $O/bench_${tag#kernel_stats}under_rocprof.json 2> $O/$tag.err
  f=$(ls -t $(find gpurun_out/prof_$tag -name "*kernel_stats.csv") | head -1)
  cp "$f" $O/$tag.csv; echo "== $tag"; head -8 $O/$tag.csv | cut -c1-160
done
bash tools/pmc_scan.sh > $O/scan_pmc.txt 2>&1
bash tools/pmc_traffic.sh > $O/scan_traffic.log 2>&1; cp gpurun_out/scan_traffic.json $O/scan_traffic.json; cat $O/scan_traffic.json | head -8
bash tools/profile_config5.sh > $O/config5_profile.log 2>&1; cp gpurun_out/c5_L3K10_kernel_stats.csv $O/config5_L3K10_kernel_stats.csv; cp gpurun_out/c5_L2K11_kernel_stats.csv $O/config5_L2K11_kernel_stats.csv
ls -la $O
