#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/d_push_timing.log 2>&1
import os, subprocess, sys, json
sys.path.insert(0, '.')
from metakssd_amd import capi
capi.Shuf.generate(11, 6, 3, 11).write('/dev/shm/L3K11.shuf')
capi.lib.mk_synth_fastq_write_mt(b'/dev/shm/big.fq', 20261002, 0, 50000000, 150, 64)
def run(tag, env_extra, args):
    env = dict(os.environ, **env_extra)
    for i in range(2):
        r = subprocess.run(['metakssd_amd/bin/metakssd', 'dist', '-L', '/dev/shm/L3K11.shuf', '-A', '-o', '/dev/shm/o', '--quiet', '--timing'] + args + ['/dev/shm/big.fq'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        for ln in r.stdout.decode().splitlines():
            if ln.startswith('{"timing"'):
                t = json.loads(ln)['timing']
                print(tag, "hip %.3f eng %.3f first %.3f last %.3f written %.3f | waitf %.3f push_call %.3f wait_call %.3f push_max %.4f first_push %.4f | H2D %.1f GB/s" % (t['hip_ready'], t['engine_ready'], t['first_push'], t['last_push'], t['written'], t['stream_wait_frame_s'], t['push_call_s'], t['wait_call_s'], t['push_call_max_s'], t['first_push_call_s'], 8.0 / (t['last_push'] - t['first_push'])), flush=True)
            elif 'engine create' in ln or 'scan cfg' in ln: pass
run('p16 c8        ', {}, ['-p', '16'])
run('p16 c8 nosdma ', {'HSA_ENABLE_SDMA': '0'}, ['-p', '16'])
run('p24 c8        ', {}, ['-p', '24'])
run('p16 c32       ', {}, ['-p', '16', '--chunk-mib', '32'])
run('p16 c8 numa1  ', {}, ['-p', '16'])
PY
cat gpurun_out/d_push_timing.log
# the same with everything bound to the GPU's NUMA node (node 1 = cpus 64-127,192-255)
numactl --hardware > gpurun_out/d_numa.txt 2>&1 || true
for i in 1 2; do taskset -c 64-127,192-255 metakssd_amd/bin/metakssd dist -L /dev/shm/L3K11.shuf -A -o /dev/shm/o --quiet --timing -p 16 /dev/shm/big.fq; done > gpurun_out/d_taskset_node1.log 2>&1
for i in 1 2; do taskset -c 0-63,128-191 metakssd_amd/bin/metakssd dist -L /dev/shm/L3K11.shuf -A -o /dev/shm/o --quiet --timing -p 16 /dev/shm/big.fq; done > gpurun_out/d_taskset_node0.log 2>&1
grep -h timing gpurun_out/d_taskset_node1.log gpurun_out/d_taskset_node0.log | python -c "
import sys, json
for ln in sys.stdin:
    t = json.loads(ln)['timing']
    print('taskset: first %.3f last %.3f written %.3f waitf %.3f push_call %.3f wait_call %.3f H2D %.1f GB/s' % (t['first_push'], t['last_push'], t['written'], t['stream_wait_frame_s'], t['push_call_s'], t['wait_call_s'], 8.0 / (t['last_push'] - t['first_push'])))
"
lscpu | grep -i numa
rm -rf /dev/shm/L3K11.shuf /dev/shm/big.fq /dev/shm/o
