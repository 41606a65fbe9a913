#!/bin/bash
# final round-2 build: config-4 regime on one GPU, the C product over several engines, the side benches and fuzz campaigns
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_session_m.sh 2>&1 | tail -40
cp gpurun_out/m_config4_one_gpu.json gpurun_out/x_config4_one_gpu.json
f=$(find gpurun_out/m_prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/x_config4_kernel_stats.csv
cp gpurun_out/m_multi_cli.log gpurun_out/x_multi_cli.jsonl
bash tools/run_side_benches.sh gpurun_out/x_side_benches.jsonl > gpurun_out/x_side.log 2>&1
cut -c1-250 gpurun_out/x_side_benches.jsonl
