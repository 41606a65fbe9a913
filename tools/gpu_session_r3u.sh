#!/bin/bash
# stage II build with the threaded upload: bench_search (build ms), tests
cd $GRAFT_REPO_ROOT
timeout 900 python tools/bench_search.py --cpu-sample 0 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_mco.py tests/test_golden.py -x -q -m gpu -k "mco or stage or search or sort or dist_r" 2>&1 | tail -3
