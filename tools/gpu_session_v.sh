#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/v_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v_pytest.log
tail -4 gpurun_out/v_pytest.log
bash tools/gpu_session_u.sh
