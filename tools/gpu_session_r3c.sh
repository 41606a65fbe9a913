#!/bin/bash
# round 3, session c: full GPU suite on the restored scan kernel + FASTA device stream + multi changes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3c_pytest.log
cat gpurun_out/r3c_pytest.log
