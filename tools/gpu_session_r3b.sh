#!/bin/bash
# round 3, session b: deferred hit test in the scan kernel -- parity first, then A/B against the previous library on this box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3b_pytest.log
cat gpurun_out/r3b_pytest.log
python tools/fuzz_parity.py --cases 1500 --seed 3001 2>&1 | tail -1
bash tools/gpu_session_ab.sh 2>&1 | tee gpurun_out/r3b_ab.txt
python tools/probe_ragged_reads.py 2>&1 | tail -8 | tee gpurun_out/r3b_ragged.txt
