#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r3e_engines.txt
timeout 1200 python -m pytest -m gpu -q --timeout=600 tests/test_golden.py -k "many_small or fasta" 2>&1 | tail -8 > gpurun_out/r3i_pytest.log
cat gpurun_out/r3i_pytest.log
bash tools/gpu_session_r3e.sh
