#!/bin/bash
# second fuzz campaign over the final build, other seeds: the engine (6 x 3000 cases), stage II / search (hand-written sort, staged upload)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/fz_campaign2.jsonl
KID=$(python -c "import bench; print(bench.kernel_source_id())")
echo "{\"kernel_source_id\": \"$KID\", \"what\": \"second fuzz campaign over this build (other seeds): engine vs oracle, stage II + search vs oracle, command line vs the compiled reference\"}" > $OUT
for seed in 3001 3002 3003 3004 3005 3006; do
  timeout 900 python tools/fuzz_parity.py --cases 3000 --seed $seed 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_parity.py','seed':$seed,'result':sys.stdin.read().strip()}))" >> $OUT
done
timeout 900 python tools/fuzz_parity.py --cases 2000 --seed 3007 --sparse 1 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_parity.py --sparse 1','seed':3007,'result':sys.stdin.read().strip()}))" >> $OUT
for seed in 3101 3102; do
  timeout 600 python tools/fuzz_search.py --seconds 150 --seed $seed 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_search.py --seconds 150','seed':$seed,'result':sys.stdin.read().strip()}))" >> $OUT
done
if [ -x oracle/_ref/metakssd ]; then
  timeout 900 python tools/fuzz_cli_vs_ref.py --cases 1500 --seed 3201 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_cli_vs_ref.py','seed':3201,'result':sys.stdin.read().strip()}))" >> $OUT
fi
cat $OUT
