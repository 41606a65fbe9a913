#!/bin/bash
# Final-tree sanity: smoke() and the driver's bench command on HEAD.
mkdir -p gpurun_out/r06j
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06j/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r06j/smoke.log
timeout 800 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06j/bench_driver.json 2> gpurun_out/r06j/bench_driver.err; echo "bench rc=$?" >> gpurun_out/r06j/smoke.log
tail -2 gpurun_out/r06j/smoke.log
