#!/bin/bash
# round 6, GPU session l: the driver's round-end GPU command on the round's last tree.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06l; mkdir -p $O
timeout 780 python3 -m pytest tests/ -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite.log
tail -3 $O/gpu_suite.log
