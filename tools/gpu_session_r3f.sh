#!/bin/bash
# round 3, session f: the whole GPU suite (per-test timeout) and then session e (bench with config 5, engines per GPU, profiles)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -40 > gpurun_out/r3f_pytest.log
cat gpurun_out/r3f_pytest.log
bash tools/gpu_session_r3e.sh
