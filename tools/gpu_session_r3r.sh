#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest -m gpu -q --timeout=600 tests/test_gpu_mco.py 2>&1 | tail -4
bash tools/gpu_session_r3q.sh 2>&1 | grep -E "mk_rs_|op" | cut -c1-200
