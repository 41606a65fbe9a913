#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --timeout=600 tests/test_gpu_parity.py -k "sparse or key_list or halves or shard" 2>&1 | tail -6
timeout 900 python -m pytest -m gpu -q --timeout=600 tests/test_gpu_fuzz.py tests/test_golden.py -k "randomized_engine or L2K11" 2>&1 | tail -4
python tools/bench_config5_variants.py 2>&1 | grep tool | grep L2K11 | cut -c1-330
