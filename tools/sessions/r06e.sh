#!/bin/bash
# round 6, GPU session e: the GPU suite under MK_POISON=0xA5 -- once whole and alone, then eight times more, four at a time (those without
# the ten tests that fill the GPU's memory on their own: tests/test_gpu_fullsize.py and the test_bench_* subprocess tests), every log kept.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06e
mkdir -p $O
MK_POISON=0xA5 timeout 200 python3 -X faulthandler -m pytest "tests/test_gpu_parity.py::test_split_queues_two_engines_in_turn" -x -q -o faulthandler_timeout=150 > $O/split_queues_under_poison.log 2>&1; echo "rc=$?" >> $O/split_queues_under_poison.log; tail -2 $O/split_queues_under_poison.log
MK_POISON=0xA5 MK_TEST_DURATIONS=$O/durations_poison_0.txt timeout 1500 python3 -m pytest tests -m gpu -q -rA --tb=long --basetemp=/tmp/pt_e0 > $O/gpu_suite_poison_0.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite_poison_0.log; tail -3 $O/gpu_suite_poison_0.log | cut -c1-200
for round in 1 2; do
  for k in 1 2 3 4; do
    n=$(( (round - 1) * 4 + k ))
    pz=0xA5; [ $k = 4 ] && pz=0x43
    ( MK_POISON=$pz timeout 1700 python3 -m pytest tests -m gpu -q -rA --tb=long --ignore=tests/test_gpu_fullsize.py -k "not test_bench_" --basetemp=/tmp/pt_e$n -p no:cacheprovider > $O/gpu_suite_poison_$n.log 2>&1; echo "suite rc=$? poison=$pz" >> $O/gpu_suite_poison_$n.log ) &
  done
  wait
done
for f in $O/gpu_suite_poison_*.log; do echo "== $f"; tail -3 $f | cut -c1-160; done
for f in $O/gpu_suite_poison_*.log; do grep -E "^(FAILED|ERROR)" $f | head -5; done
gzip -9 $O/gpu_suite_poison_*.log
ls gpurun_out | grep fail_ | head
du -sh gpurun_out
