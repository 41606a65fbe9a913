#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/q_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/q_pytest.log
tail -3 gpurun_out/q_pytest.log
timeout 900 python bench.py > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/q_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, 'frac', round(d['roofline']['frac'], 4), d['phases_ms_per_step'])
print('t_stream', {k: d['t_stream'].get(k) for k in ('gbases_s', 'h2d_gb_s', 'seconds')})
print('t_e2e', {k: d['t_e2e'].get(k) for k in ('gbases_s', 'seconds', 'init_s', 'gbases_s_wall', 'wall_s', 'sketch_equals_resident_run')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_equals_reference_multiset'])
PY
bash tools/pmc_traffic.sh > gpurun_out/q_traffic.log 2>&1; tail -1 gpurun_out/q_traffic.log | head -c 400; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/q_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_prof -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-legs > gpurun_out/q_prof.log 2>&1
find gpurun_out/q_prof -name "*kernel_stats.csv" | head -1 | xargs head -6
rm -rf gpurun_out/pmc_*
bash tools/pmc_scan.sh > gpurun_out/q_pmc.log 2>&1
python3 tools/pmc_summary.py > gpurun_out/q_pmc_summary.txt 2>&1; cat gpurun_out/q_pmc_summary.txt | head -30
