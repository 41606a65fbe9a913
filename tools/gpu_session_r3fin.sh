#!/bin/bash
# after the host-side change: command line vs the compiled reference (FASTQ files through the mapped stream), then the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/fuzz_cli_vs_ref.py --cases 1500 --seed 4001 2>&1 | tail -1
timeout 1500 python bench.py > gpurun_out/r3fin_bench.json 2> gpurun_out/r3fin_bench.err
tail -c 600 gpurun_out/r3fin_bench.json
