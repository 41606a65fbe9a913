#!/bin/bash
# round 3, session h: whole GPU suite again (after the fix of the stream-buffer reserve), bench with config 5, config-5 kernel statistics
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -40 > gpurun_out/r3h_pytest.log
cat gpurun_out/r3h_pytest.log
bash tools/gpu_session_r3e.sh
