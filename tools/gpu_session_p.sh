#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_*
bash tools/pmc_scan.sh > gpurun_out/p_pmc.log 2>&1
python3 tools/pmc_summary.py > gpurun_out/p_pmc_summary.txt 2>&1
cat gpurun_out/p_pmc_summary.txt
