#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/i_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/i_pytest.log
tail -2 gpurun_out/i_pytest.log
python - <<'PY' > gpurun_out/i_push_timing.log 2>&1
import os, subprocess, sys, json, time
sys.path.insert(0, '.')
from metakssd_amd import capi
capi.Shuf.generate(11, 6, 3, 11).write('/dev/shm/L3K11.shuf')
capi.lib.mk_synth_fastq_write_mt(b'/dev/shm/big.fq', 20261002, 0, 50000000, 150, 64)
def run(tag, env_extra, args, pre=[], reps=3, pause=0.0):
    env = dict(os.environ, **env_extra)
    for i in range(reps):
        time.sleep(pause)
        r = subprocess.run(pre + ['metakssd_amd/bin/metakssd', 'dist', '-L', '/dev/shm/L3K11.shuf', '-A', '-o', '/dev/shm/o', '--quiet', '--timing'] + args + ['/dev/shm/big.fq'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        for ln in r.stdout.decode().splitlines():
            if ln.startswith('{"timing"'):
                t = json.loads(ln)['timing']
                print(tag, "hip %.3f eng %.3f first %.3f last %.3f written %.3f | setup %.3f waitf %.3f push_call %.3f wait_call %.3f first_push %.4f | H2D %.1f GB/s | written-hip %.3f -> %.1f Gbases/s" % (t['hip_ready'], t['engine_ready'], t['first_push'], t['last_push'], t['written'], t['stream_setup_s'], t['stream_wait_frame_s'], t['push_call_s'], t['wait_call_s'], t['first_push_call_s'], 8.0 / (t['last_push'] - t['first_push']), t['written'] - t['hip_ready'], 7.5 / (t['written'] - t['hip_ready'])), flush=True)
            elif 'engine create' in ln: print('   ', ln)
run('warm          ', {}, ['-p', '16'], reps=1)
run('p16 c8        ', {}, ['-p', '16'])
run('p16 c8 pause1 ', {}, ['-p', '16'], pause=1.0)
run('p16 c8 ticks  ', {'LD_LIBRARY_PATH': os.path.abspath('metakssd_amd/lib_tuning'), 'MK_DEBUG': '1'}, ['-p', '16'], reps=2, pause=1.0)
run('p16 c16 pause1', {}, ['-p', '16', '--chunk-mib', '16'], pause=1.0)
run('p20 c8 pause1 ', {}, ['-p', '20'], pause=1.0)
PY
cat gpurun_out/i_push_timing.log
rm -rf /dev/shm/L3K11.shuf /dev/shm/big.fq /dev/shm/o
