#!/bin/bash
# SQ counters of the resolve kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_*
bash tools/pmc_scan.sh > gpurun_out/s_pmc.log 2>&1
python3 tools/pmc_summary.py 356250 resolve_kernel > gpurun_out/s_pmc_resolve.txt 2>&1; cat gpurun_out/s_pmc_resolve.txt
