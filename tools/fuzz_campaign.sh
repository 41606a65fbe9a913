#!/bin/bash
# fuzz campaigns over the final build (engine vs oracle, CLI vs compiled reference): totals as JSON lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/fz_campaign.jsonl
# every line carries the id of the kernel source it ran on (what bench.py reports as roofline.kernel_source_id)
KID=$(python -c "import bench; print(bench.kernel_source_id())")
echo "{\"kernel_source_id\": \"$KID\", \"what\": \"fuzz campaign over this build: engine vs oracle, command line vs the compiled reference, framers\"}" > $OUT
for seed in 2001 2002 2003; do
  python tools/fuzz_parity.py --cases 3000 --seed $seed 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_parity.py','seed':$seed,'result':sys.stdin.read().strip()}))" >> $OUT
done
python tools/fuzz_parity.py --cases 600 --seed 2004 --big --big-rate 0.3 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_parity.py --big --big-rate 0.3','seed':2004,'result':sys.stdin.read().strip()}))" >> $OUT
python tools/fuzz_parity.py --cases 1500 --seed 2005 --sparse 0 --front-bits 5 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_parity.py --sparse 0 --front-bits 5','seed':2005,'result':sys.stdin.read().strip()}))" >> $OUT
python tools/fuzz_parity.py --cases 1500 --seed 2006 --sparse 1 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_parity.py --sparse 1','seed':2006,'result':sys.stdin.read().strip()}))" >> $OUT
if [ -x oracle/_ref/metakssd ]; then
  python tools/fuzz_cli_vs_ref.py --cases 1000 --seed 2007 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_cli_vs_ref.py','seed':2007,'result':sys.stdin.read().strip()}))" >> $OUT
fi
python tools/fuzz_framing.py --cases 2000 --seed 2008 2>&1 | tail -1 | python -c "import sys,json; print(json.dumps({'tool':'tools/fuzz_framing.py','seed':2008,'result':sys.stdin.read().strip()}))" >> $OUT
cat $OUT
