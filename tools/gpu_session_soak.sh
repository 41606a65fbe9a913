#!/bin/bash
# soak: the L2K11 command-line tests (21 GB engines created and torn down per process) over and over, a timeout per test
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${SOAK_ROUNDS:-12}); do
  timeout 600 python -m pytest tests/test_golden.py -x -q -m gpu -k "L2K11" --timeout=150 2>&1 | tail -1
done
