#!/bin/bash
# every number DESIGN.md quotes outside bench.py, as JSON lines in one file (copied into profiles/ by whoever ran it):
#   usage: bash tools/run_side_benches.sh gpurun_out/r02_side_benches.jsonl
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/side_benches.jsonl}
: > $OUT
run() { echo "## $*" >&2; "$@" 2>&1 | grep -E '^\{' >> $OUT; }
GENOMES=${GENOMES:-512} THREADS=${THREADS:-16} run python tools/bench_config5.py
run python tools/bench_set.py
run python tools/bench_join.py
run python tools/bench_search.py
# randomized differential campaigns (engine vs oracle, CLI vs compiled reference): totals as JSON
for seed in 101 102; do
  python tools/fuzz_parity.py --cases 1500 --seed $seed 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_parity.py','seed':$seed,'result':sys.stdin.read().strip()}))" >> $OUT
done
python tools/fuzz_setop.py --cases 1500 --seed 103 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_setop.py','seed':103,'result':sys.stdin.read().strip()}))" >> $OUT
python tools/fuzz_search.py --seconds 60 --seed 104 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_search.py','seed':104,'result':sys.stdin.read().strip()}))" >> $OUT
if [ -x oracle/_ref/metakssd ]; then
  python tools/fuzz_cli_vs_ref.py --cases 400 --seed 105 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_cli_vs_ref.py','seed':105,'result':sys.stdin.read().strip()}))" >> $OUT
fi
cat $OUT
