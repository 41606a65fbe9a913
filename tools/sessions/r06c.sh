#!/bin/bash
# round 6, GPU session c: where MK_POISON's time goes (per-test durations, plain and poisoned, on the golden + parity files), the profiler
# evidence of the round (tools/profile_round.sh), config 5's per-batch timeline and variants taking turns, the new tests.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06c
mkdir -p $O
# 1. the tests that failed / are new
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -rA --tb=long -k "refuses_more_engines or bare_two_ranks or default_flow_one_gpu or without_kmer_in_a_component or launches_its_own_ranks or two_rank_flow" > $O/new_tests.log 2>&1; echo "rc=$?" >> $O/new_tests.log
tail -4 $O/new_tests.log
# 2. per-test durations: golden file plain, then under poison (bounded)
MK_TEST_DURATIONS=$O/durations_golden_plain.txt timeout 600 python3 -m pytest tests/test_golden.py -m gpu -q > $O/golden_plain.log 2>&1; tail -1 $O/golden_plain.log
MK_POISON=0xA5 MK_TEST_DURATIONS=$O/durations_golden_poison.txt timeout 900 python3 -m pytest tests/test_golden.py -m gpu -q > $O/golden_poison.log 2>&1; tail -1 $O/golden_poison.log
MK_POISON=0xA5 MK_TEST_DURATIONS=$O/durations_parity_poison.txt timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_setop.py tests/test_gpu_mco.py -m gpu -q > $O/parity_poison.log 2>&1; tail -1 $O/parity_poison.log
sort -rn $O/durations_golden_poison.txt | head -8; sort -rn $O/durations_parity_poison.txt | head -12
# 3. config 5: the per-batch timeline and variants taking turns
python3 tools/trace_config5.py --geometry L2K11 > $O/trace_c5_L2K11.json 2> $O/trace_c5_L2K11.err; cut -c1-1800 $O/trace_c5_L2K11.json
MK_BATCH_TAB_BITS=16 python3 tools/trace_config5.py --geometry L2K11 > $O/trace_c5_L2K11_tb16.json 2> $O/trace_c5_L2K11_tb16.err; cut -c1-1200 $O/trace_c5_L2K11_tb16.json
python3 tools/trace_config5.py --geometry L3K10 > $O/trace_c5_L3K10.json 2> $O/trace_c5_L3K10.err; cut -c1-1200 $O/trace_c5_L3K10.json
python3 tools/ab_config5.py --geometry L2K11 --variant default --variant "MK_BATCH_TAB_BITS=16" --variant "MK_BATCH_TAB_BITS=15" --variant "flags:--batch-files 64 --batch-mib 256;MK_BATCH_TAB_BITS=16" > $O/ab_c5_L2K11.jsonl 2> $O/ab_c5_L2K11.err; cat $O/ab_c5_L2K11.jsonl
# 4. the round's profiler evidence
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -25 $O/profile_round.log | cut -c1-200
du -sh gpurun_out
