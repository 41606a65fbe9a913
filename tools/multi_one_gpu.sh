#!/bin/bash
# the N-rank flows on ONE GPU (the ranks share it; gloo / device copies instead of RCCL: phases and equality, nothing about links):
# both merges at BASELINE config 3 and config 4 sizes, process-per-GPU harness + the C product's libmetakssd_multi.so (--inproc-multi)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_multi_one_gpu.jsonl; : > $O
for total in 50000000 500000000; do
  for merge in gather slices; do
    python3 bench.py --gpus 8 --total-reads $total --steps 5 --warmup 1 --no-host-legs --merge $merge --verify --inproc-multi 2> gpurun_out/multi_${total}_$merge.err | grep '^{' >> $O
    tail -1 $O | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
im=d.get('inproc_multi',{})
print('$total $merge: ms/step %.2f tail %.2f ms phases %s equal %s | inproc: ms/step %s tail %s phases %s equal %s' % (d['ms_per_step'], d.get('rank0_tail_ms',0), {k:round(v,2) for k,v in d.get('rank0_tail_phases_ms',{}).items()}, d.get('merged_equals_single_engine'), im.get('ms_per_step'), im.get('tail_ms'), {k:round(v,2) for k,v in im.get('tail_phases_ms',{}).items()}, im.get('equals_process_per_gpu_sketch')))"
  done
done
