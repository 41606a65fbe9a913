#!/bin/bash
# round 6, GPU session b: the GPU suite plain and under MK_POISON (full logs kept), the pipeline fuzzer under poison, the bench lines.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06b
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q -rA --tb=long > $O/gpu_suite_plain.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite_plain.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_steps20.json 2> $O/bench_driver_steps20.err
python3 tools/fuzz_pipeline.py --cases 1000 --workers 32 --l2k11 0.08 --poison 0xA5 --seed 62 > $O/fuzz_pipeline_poisonA5.json 2> $O/fuzz_pipeline_poisonA5.err
MK_POISON=0xA5 timeout 2400 python3 -m pytest tests -m gpu -q -rA --tb=long > $O/gpu_suite_poisonA5.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite_poisonA5.log
gzip -9 $O/gpu_suite_plain.log $O/gpu_suite_poisonA5.log
for f in $O/*.json; do echo "== $f"; cut -c1-1500 $f; done
zcat $O/gpu_suite_plain.log.gz | tail -15 | cut -c1-300
zcat $O/gpu_suite_poisonA5.log.gz | tail -8 | cut -c1-300
du -sh gpurun_out
