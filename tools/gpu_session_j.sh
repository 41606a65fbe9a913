#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_golden.py -m gpu -x -q -k "dropin" > gpurun_out/j_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/j_pytest.log
tail -3 gpurun_out/j_pytest.log
python tools/probe_direct_scan.py > gpurun_out/j_direct_scan.log 2>&1; tail -8 gpurun_out/j_direct_scan.log
python - <<'PY' > gpurun_out/j_push_timing.log 2>&1
import os, subprocess, sys, json, time
sys.path.insert(0, '.')
from metakssd_amd import capi
capi.Shuf.generate(11, 6, 3, 11).write('/dev/shm/L3K11.shuf')
capi.lib.mk_synth_fastq_write_mt(b'/dev/shm/big.fq', 20261002, 0, 50000000, 150, 64)
def run(tag, env_extra, args, pre=[], reps=3, pause=1.0, exe='metakssd_amd/bin/metakssd'):
    env = dict(os.environ, **env_extra)
    for i in range(reps):
        time.sleep(pause)
        t0 = time.perf_counter()
        r = subprocess.run(pre + [exe, 'dist', '-L', '/dev/shm/L3K11.shuf', '-A', '-o', '/dev/shm/o'] + args + ['/dev/shm/big.fq'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        wall = time.perf_counter() - t0
        got = False
        for ln in r.stdout.decode().splitlines():
            if ln.startswith('{"timing"'):
                got = True
                t = json.loads(ln)['timing']
                print(tag, "wall %.3f hip %.3f eng %.3f first %.3f last %.3f written %.3f | begin %.4f setup %.3f waitf %.3f push_call %.3f wait_call %.3f first_push %.4f | H2D %.1f GB/s | written-hip %.3f -> %.1f Gbases/s" % (wall, t['hip_ready'], t['engine_ready'], t['first_push'], t['last_push'], t['written'], t['begin_s'], t['stream_setup_s'], t['stream_wait_frame_s'], t['push_call_s'], t['wait_call_s'], t['first_push_call_s'], 8.0 / (t['last_push'] - t['first_push']), t['written'] - t['hip_ready'], 7.5 / (t['written'] - t['hip_ready'])), flush=True)
        if not got: print(tag, 'wall %.3f rc %d' % (wall, r.returncode), r.stdout.decode()[-200:].replace('\n', ' | '))
run('warm          ', {}, ['--quiet', '--timing', '-p', '16'], reps=1)
run('p20 c8        ', {}, ['--quiet', '--timing', '-p', '20'])
run('p20 c16       ', {}, ['--quiet', '--timing', '-p', '20', '--chunk-mib', '16'])
run('ref+dropin p20', {}, ['-p', '20'], exe='oracle/_ref_hip/metakssd', reps=2)
import hashlib
print('product  ', hashlib.sha256(open('/dev/shm/o/combco.0','rb').read()).hexdigest()[:16], hashlib.sha256(open('/dev/shm/o/combco.0.a','rb').read()).hexdigest()[:16])
PY
cat gpurun_out/j_push_timing.log
rm -rf /dev/shm/L3K11.shuf /dev/shm/big.fq /dev/shm/o
