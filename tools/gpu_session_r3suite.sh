#!/bin/bash
# the whole GPU suite on the tree as it is
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -15 > gpurun_out/r3suite_pytest.log
cat gpurun_out/r3suite_pytest.log
