#!/bin/bash
# round 6, GPU session a: the golden composite pipelines again and again (with and without MK_POISON), the pipeline fuzzer,
# one full GPU suite under MK_POISON, one default bench line.  Full logs are kept (never piped through tail).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06a
mkdir -p $O
python3 tools/fuzz_pipeline.py --golden composite_mix_L2K11 --times 200 --workers 4 --poison 0xA5 > $O/golden_L2K11_poisonA5.json 2> $O/golden_L2K11_poisonA5.err
python3 tools/fuzz_pipeline.py --golden composite_mix_L2K11 --times 200 --workers 8 > $O/golden_L2K11_plain_w8.json 2> $O/golden_L2K11_plain_w8.err
python3 tools/fuzz_pipeline.py --golden composite_mix_L2K11 --times 100 --workers 4 --poison 0x43 > $O/golden_L2K11_poison43.json 2> $O/golden_L2K11_poison43.err
python3 tools/fuzz_pipeline.py --cases 600 --workers 24 --poison 0xA5 --seed 61 > $O/fuzz_pipeline_poisonA5.json 2> $O/fuzz_pipeline_poisonA5.err
MK_POISON=0xA5 timeout 1500 python3 -m pytest tests -m gpu -q -rA --tb=long > $O/gpu_suite_poisonA5.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite_poisonA5.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?" >> $O/bench_default.err
cat $O/*.json | cut -c1-600
tail -5 $O/gpu_suite_poisonA5.log
