#!/bin/bash
# round 6, GPU session g: the driver's sequence once more on HEAD (bench.py's ceiling measurement changed after session f): smoke, the driver's
# command, and the bare two-rank line on this one-GPU box.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06g; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_steps20.json 2> $O/bench_driver_steps20.err; echo "rc=$?"
python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err; echo "rc=$?"
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r06g/bench_driver_steps20.json'))
print('N=1', round(j['value'],1), round(j['ms_per_step'],4), 'frac', round(j['roofline']['frac'],4), 't_e2e', round(j['t_e2e']['gbases_s'],1), 'ceiling', round(j['t_e2e']['ceiling']['gbases_s'],1), [r['gbases_s'] for r in j['t_e2e']['all_runs']])
l=[x for x in open('gpurun_out/r06g/bench_two_ranks_one_gpu.json') if x.startswith('{')][-1]
j=json.loads(l)
print('N=2', round(j['value'],1), round(j['ms_per_step'],3), j['scaling'], j['config']['parallelism'][:60], {k:j['same_workload_one_gpu'].get(k) for k in ('ms_per_step','speedup','efficiency','sketch_equals_merged')},
      {k:j['inproc_multi'].get(k) for k in ('ms_per_step','transport','merge','equals_process_per_gpu_sketch')}, j['distributed']['ranks_seen'])
PY
