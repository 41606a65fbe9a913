#!/bin/bash
# round-2 "c" artifacts: default bench line, kernel stats, TCC traffic, SQ counters (scan + resolve)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_traffic.sh > gpurun_out/w_traffic.log 2>&1; tail -1 gpurun_out/w_traffic.log | head -c 300; echo
cp gpurun_out/scan_traffic.json profiles/scan_traffic.json   # so that the bench line below reports it (same kernel source)
timeout 900 python bench.py > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/w_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, 'roofline', d['roofline'], d['phases_ms_per_step'])
print('t_stream', {k: d['t_stream'].get(k) for k in ('gbases_s', 'h2d_gb_s', 'seconds')})
print('t_e2e', {k: d['t_e2e'].get(k) for k in ('gbases_s', 'seconds', 'init_s', 'gbases_s_wall', 'wall_s', 'sketch_equals_resident_run')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_equals_reference_multiset'])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/w_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w_prof -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-legs > gpurun_out/w_prof.log 2>&1
f=$(find gpurun_out/w_prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/w_kernel_stats.csv; head -14 $f | cut -c1-150
rm -rf gpurun_out/pmc_*
bash tools/pmc_scan.sh > gpurun_out/w_pmc.log 2>&1
python3 tools/pmc_summary.py > gpurun_out/w_pmc_scan.txt 2>&1; head -30 gpurun_out/w_pmc_scan.txt
python3 tools/pmc_summary.py 356250 resolve_kernel > gpurun_out/w_pmc_resolve.txt 2>&1
