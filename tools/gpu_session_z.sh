#!/bin/bash
# candidate buffers: 8192 records per scan wave (a pitch of 128 KiB between the waves' buffers) against pitches that are not powers of two
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for c in 8192 8208 8256 8448 9000; do
python bench.py --cand-cap $c --steps 60 --warmup 3 --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cand_cap $c', {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, round(d['ms_per_step'],3))"
done
done
