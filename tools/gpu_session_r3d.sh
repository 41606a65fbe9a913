#!/bin/bash
# round 3, session d: the new paths first, every test under a timeout (a hang gives a traceback, not a lost box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3d_pytest.log
: > $L
run() { echo "=== $*" >> $L; timeout 1500 python -m pytest -m gpu -q --timeout=400 -x "$@" 2>&1 | tail -25 >> $L; }
timeout 600 python tools/diag_composite_hang.py > gpurun_out/r3d_diag.log 2>&1; tail -60 gpurun_out/r3d_diag.log
run tests/test_golden.py -k "composite_mix_L2K11"
run tests/test_gpu_parity.py -k "fasta or stream"
run tests/test_gpu_parity.py -k "not fasta and not stream"
run tests/test_gpu_fuzz.py
run tests/test_golden.py -k "fasta or many_small or devices or several_engines"
cat $L
