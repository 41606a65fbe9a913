#!/bin/bash
# resolve kernel: fixed cost (small inputs) against the 50 M-read launch
cd $GRAFT_REPO_ROOT
for n in 100000 1000000 5000000 50000000; do
python bench.py --reads-per-gpu $n --steps 50 --warmup 3 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('reads $n', {k: round(v,4) for k,v in d['phases_ms_per_step'].items()}, 'ms', round(d['ms_per_step'],3))"
done
