#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -q --timeout=300 -x "tests/test_gpu_parity.py::test_fasta_device_stream_equals_oracle" 2>&1 | tail -60 > gpurun_out/r3g_a.log
timeout 900 python -m pytest -m gpu -q --timeout=300 -x "tests/test_gpu_parity.py::test_fastq_set_min_occurrence" 2>&1 | tail -60 > gpurun_out/r3g_b.log
timeout 900 python tools/fuzz_parity.py --cases 250 --seed 12 --sparse 1 2>&1 | tail -30 > gpurun_out/r3g_c.log
cat gpurun_out/r3g_a.log gpurun_out/r3g_b.log gpurun_out/r3g_c.log
