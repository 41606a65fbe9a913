#!/bin/bash
# the FASTQ command line N times in a row per variant (the bench's t_e2e leg saw its LAST run's HIP start-up take 0.22-0.25 s): does a run's
# start-up depend on how many runs came before it, and on whether the row pool asks for huge pages?
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json, os, subprocess, sys, time
sys.path.insert(0, '.')
from metakssd_amd import capi
d = '/dev/shm/mk_seq'; os.makedirs(d, exist_ok=True)
fq, sp = d + '/reads.fq', d + '/L3K11.shuf'
capi.Shuf.generate(11, 6, 3, 11).write(sp)
assert capi.lib.mk_synth_fastq_write_mt(fq.encode(), 20261002, 0, 50_000_000, 150, 64) == 0
cli = 'metakssd_amd/bin/metakssd'
def run(tag, env, sleep):
    time.sleep(sleep)
    subprocess.run(['rm', '-rf', d + '/out'])
    t0 = time.monotonic()
    r = subprocess.run([cli, 'dist', '-L', sp, '-A', '-o', d + '/out', '--quiet', '--timing', fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    wall = time.monotonic() - t0
    tm = [json.loads(l)['timing'] for l in r.stdout.decode().splitlines() if l.startswith('{"timing"')][0]
    print('%-10s sleep %.1f rc %d wall %.3f written %.3f hip %.3f eng %.3f last_push %.3f wait_frame %.3f' % (tag, sleep, r.returncode, wall, tm['written'], tm['hip_ready'], tm['engine_ready'], tm['last_push'], tm['stream_wait_frame_s']), flush=True)
run('warm', {}, 1.0)
for block in range(2):
    for tag, env in (('thp', {}), ('no_thp', {'MK_NO_THP': '1'})):
        for i in range(8):
            run(tag, env, 2.5)
for i in range(6):
    run('thp', {}, 0.3)
subprocess.run(['rm', '-rf', d])
PY
