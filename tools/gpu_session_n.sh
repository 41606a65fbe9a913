#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_golden.py tests/test_gpu_mco.py -m gpu -x -q -k "sixteen or mco or several or variants" > gpurun_out/n_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/n_pytest.log
tail -4 gpurun_out/n_pytest.log
timeout 1500 bash tools/run_side_benches.sh gpurun_out/n_side_benches.jsonl > gpurun_out/n_side.log 2>&1
tail -25 gpurun_out/n_side_benches.jsonl | cut -c1-400
