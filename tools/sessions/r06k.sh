#!/bin/bash
# round 6, GPU session k: one more campaign of random pipelines on the round's last tree, the other poison byte, twice the share of L2K11.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06k; mkdir -p $O
timeout 800 python3 tools/fuzz_pipeline.py --cases 1200 --seed 64 --workers 32 --l2k11 0.3 --poison 0x43 --keep $O/fail > $O/fuzz_pipeline.log 2> $O/fuzz_pipeline.err; echo "rc=$?" >> $O/fuzz_pipeline.log
tail -4 $O/fuzz_pipeline.log | cut -c1-600
