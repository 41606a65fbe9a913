#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --timeout=600 tests/test_golden.py -k "many_small" 2>&1 | tail -3
python tools/bench_config5_variants.py 2>&1 | grep tool | cut -c1-330
