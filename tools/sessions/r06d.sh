#!/bin/bash
# round 6, GPU session d: what test_split_queues_two_engines_in_turn does under MK_POISON (it did not return in sessions b and c); then the GPU suite
# (plain) with durations on the tree with the smaller batch tables and the one-pass predicates; bench lines; config 5 again.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06d
mkdir -p $O
# 1. the hang: python stacks after 40 s, what the GPU does meanwhile, and which half of the hook it needs
( sleep 45; rocm-smi --showuse --showmemuse > $O/hang_gpu_use.txt 2>&1 ) &
MK_POISON=0xA5 timeout 90 python3 -X faulthandler -m pytest "tests/test_gpu_parity.py::test_split_queues_two_engines_in_turn" -x -q -o faulthandler_timeout=40 > $O/hang_poison_full.log 2>&1; echo "rc=$?" >> $O/hang_poison_full.log
wait
MK_POISON=0xA5 MK_POISON_SCRATCH=0 timeout 90 python3 -X faulthandler -m pytest "tests/test_gpu_parity.py::test_split_queues_two_engines_in_turn" -x -q -o faulthandler_timeout=40 > $O/hang_poison_alloc_only.log 2>&1; echo "rc=$?" >> $O/hang_poison_alloc_only.log
tail -3 $O/hang_poison_alloc_only.log; grep -n "File \"/root/repo\|capi.py\|Thread\|rc=" $O/hang_poison_full.log | head -30; cat $O/hang_gpu_use.txt | head -20
# 2. the suite, plain, with durations
MK_TEST_DURATIONS=$O/durations_suite_plain.txt timeout 1500 python3 -m pytest tests -m gpu -q -rA --tb=long > $O/gpu_suite_plain.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite_plain.log
tail -6 $O/gpu_suite_plain.log | cut -c1-300
# 3. bench lines
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_steps20.json 2> $O/bench_driver_steps20.err
python3 tools/ab_config5.py --geometry L2K11 --variant default --rounds 2 > $O/ab_c5_L2K11.jsonl 2> $O/ab_c5_L2K11.err; cat $O/ab_c5_L2K11.jsonl
python3 tools/trace_config5.py --geometry L2K11 > $O/trace_c5_L2K11.json 2> $O/trace_c5_L2K11.err; cut -c1-1500 $O/trace_c5_L2K11.json
gzip -9 $O/gpu_suite_plain.log
du -sh gpurun_out
