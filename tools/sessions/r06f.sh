#!/bin/bash
# round 6, GPU session f (the final tree): four suites at a time under MK_POISON once more (the tests' copy race and port collisions fixed, the tests that
# write 32 GiB indexes left to the serial run), the probe-count ablation, the round's profiler evidence, the suite alone, the bench lines, the reference's
# stage II on the next_rows directory, a fuzz campaign under poison.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06f
mkdir -p $O
for k in 1 2 3 4; do
  pz=0xA5; [ $k = 4 ] && pz=0x43
  ( MK_POISON=$pz timeout 1500 python3 -m pytest tests -m gpu -q -rA --tb=long --ignore=tests/test_gpu_fullsize.py -k "not test_bench_ and not stage2_search_end_to_end and not sixteen_component_database_end_to_end" --basetemp=/tmp/pt_f$k -p no:cacheprovider > $O/gpu_suite_poison_$k.log 2>&1; echo "suite rc=$? poison=$pz" >> $O/gpu_suite_poison_$k.log ) &
done
wait
for f in $O/gpu_suite_poison_*.log; do echo "== $f"; tail -2 $f | cut -c1-160; grep -E "^(FAILED|ERROR)" $f | head -5; done
rm -rf /tmp/pt_f1 /tmp/pt_f2 /tmp/pt_f3 /tmp/pt_f4
bash tools/sessions/r06_probe_ablation.sh > $O/probe_ablation.log 2>&1; tail -7 $O/probe_ablation.log
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -22 $O/profile_round.log | cut -c1-180
MK_TEST_DURATIONS=$O/durations_suite.txt timeout 1500 python3 -m pytest tests -m gpu -q -rA --tb=long > $O/gpu_suite_plain.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite_plain.log; tail -4 $O/gpu_suite_plain.log | cut -c1-200
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_steps20.json 2> $O/bench_driver_steps20.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_steps20_run2.json 2> $O/bench_driver_steps20_run2.err
python3 tools/next_rows_reference_stage2.py --timeout 900 > $O/next_rows_reference_stage2.json 2> $O/next_rows_reference_stage2.err; cut -c1-900 $O/next_rows_reference_stage2.json
MK_POISON=0xA5 python3 tools/fuzz_parity.py --cases 1500 --seed 606 > $O/fuzz_parity_poison.log 2>&1; tail -1 $O/fuzz_parity_poison.log
MK_POISON=0xA5 python3 tools/fuzz_cli_vs_ref.py --cases 300 --seed 66 > $O/fuzz_cli_vs_ref_poison.log 2>&1; tail -1 $O/fuzz_cli_vs_ref_poison.log
gzip -9 $O/gpu_suite_poison_*.log $O/gpu_suite_plain.log
python3 - <<'PY'
import json
for f in ('bench_default','bench_driver_steps20','bench_driver_steps20_run2'):
    try:
        j=json.load(open('gpurun_out/r06f/%s.json'%f))
        print(f, round(j['value'],1), round(j['ms_per_step'],4), 'scan', round(j['roofline']['avg_launch_ms'],4), 'frac', round(j['roofline']['frac'],4), 'id', j['roofline']['kernel_source_id'],
              't_e2e', round(j['t_e2e']['gbases_s'],1), 'c5', [round(j['config5'][g]['roofline']['frac'],3) for g in ('L3K10','L2K11')])
    except Exception as ex:
        print(f, 'unreadable', ex)
PY
du -sh gpurun_out
