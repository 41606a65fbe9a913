#!/bin/bash
# round 6, GPU session h: the whole GPU suite alone under MK_POISON on the final tree; 2 000 more random MarkerDB pipelines under poison against oracle +
# compiled reference; the engine fuzz campaign (tools/fuzz_campaign.sh) under poison.
cd "$(dirname "$0")/../.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r06h; mkdir -p $O
MK_POISON=0xA5 timeout 1500 python3 -m pytest tests -m gpu -q -rA --tb=long > $O/gpu_suite_poison_alone.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite_poison_alone.log; tail -3 $O/gpu_suite_poison_alone.log | cut -c1-200
python3 tools/fuzz_pipeline.py --cases 2000 --workers 32 --l2k11 0.06 --poison 0xA5 --seed 63 > $O/fuzz_pipeline_poisonA5.json 2> $O/fuzz_pipeline_poisonA5.err; cut -c1-700 $O/fuzz_pipeline_poisonA5.json
MK_POISON=0xA5 bash tools/fuzz_campaign.sh > $O/fuzz_campaign.log 2>&1; cp gpurun_out/fz_campaign.jsonl $O/fuzz_campaign_poison.jsonl; cat $O/fuzz_campaign_poison.jsonl | cut -c1-260
MK_POISON=0xA5 python3 tools/fuzz_setop.py --cases 600 --seed 68 > $O/fuzz_setop.log 2>&1; tail -1 $O/fuzz_setop.log
MK_POISON=0xA5 python3 tools/fuzz_search.py --seconds 90 --seed 69 > $O/fuzz_search.log 2>&1; tail -1 $O/fuzz_search.log
gzip -9 $O/gpu_suite_poison_alone.log
