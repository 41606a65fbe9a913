#!/bin/bash
# BASELINE config 4's table regime on one GPU with the final build: 500 M reads whole vs 8 shards merged, + kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python bench.py --total-reads 500000000 --steps 5 --warmup 1 --verify --no-cpu-baseline --no-host-legs > gpurun_out/c4_one_gpu.json 2> gpurun_out/c4.err; echo "config4 rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c4_one_gpu.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'scaling')}, d['config']['distinct_keys'], d['config']['table_load'], d.get('merged_equals_single_engine'), d['phases_ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c4_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4_prof -- python3 bench.py --total-reads 500000000 --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs > gpurun_out/c4_prof.log 2>&1
f=$(find gpurun_out/c4_prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/c4_kernel_stats.csv; head -9 $f | cut -c1-160
