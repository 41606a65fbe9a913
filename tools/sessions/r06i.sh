#!/bin/bash
# round 6, GPU session i: config 4's N = 1 point as a line of its own, and the N = 8 line on this one-GPU box (ranks on GPU 0 over gloo: a debug
# transport, said in the line) with what it carries without flags.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06i; mkdir -p $O
python3 bench.py --gpus 1 --total-reads 500000000 --steps 20 --warmup 3 > $O/bench_config4_one_gpu.json 2> $O/bench_config4_one_gpu.err; echo "rc=$?"
python3 bench.py --gpus 8 --steps 10 --warmup 2 --no-host-legs > $O/bench_eight_ranks_one_gpu.json 2> $O/bench_eight_ranks_one_gpu.err; echo "rc=$?"
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r06i/bench_config4_one_gpu.json'))
print('N=1 config4', round(j['value'],1), round(j['ms_per_step'],3), j['scaling'], j['config']['workload'][:90], j['config']['distinct_keys'], round(j['config']['table_load'],3), 'scan', round(j['roofline']['avg_launch_ms'],3), round(j['roofline']['frac'],3))
l=[x for x in open('gpurun_out/r06i/bench_eight_ranks_one_gpu.json') if x.startswith('{')][-1]
j=json.loads(l)
print('N=8', round(j['value'],1), round(j['ms_per_step'],3), j['merge'], {k:j['same_workload_one_gpu'].get(k) for k in ('ms_per_step','speedup','efficiency','sketch_equals_merged')},
      {k:j['inproc_multi'].get(k) for k in ('ms_per_step','transport','merge','equals_process_per_gpu_sketch','tail_ms')}, j['distributed']['ranks_seen'], j['config']['distinct_keys'])
PY
