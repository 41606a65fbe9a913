#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest -m gpu -q --timeout=600 tests/test_gpu_mco.py 2>&1 | tail -8
timeout 1200 python -m pytest -m gpu -q --timeout=900 tests/test_golden.py -k "stage2 or search or sixteen" 2>&1 | tail -4
timeout 600 python tools/bench_search.py --cpu-sample 0 2>&1 | tail -6 | cut -c1-400
