#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/g_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g_pytest.log
tail -3 gpurun_out/g_pytest.log
python - <<'PY' > gpurun_out/g_push_timing.log 2>&1
import os, subprocess, sys, json
sys.path.insert(0, '.')
from metakssd_amd import capi
capi.Shuf.generate(11, 6, 3, 11).write('/dev/shm/L3K11.shuf')
capi.lib.mk_synth_fastq_write_mt(b'/dev/shm/big.fq', 20261002, 0, 50000000, 150, 64)
def run(tag, env_extra, args, pre=[], reps=3):
    env = dict(os.environ, **env_extra)
    for i in range(reps):
        r = subprocess.run(pre + ['metakssd_amd/bin/metakssd', 'dist', '-L', '/dev/shm/L3K11.shuf', '-A', '-o', '/dev/shm/o', '--quiet', '--timing'] + args + ['/dev/shm/big.fq'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        for ln in r.stdout.decode().splitlines():
            if ln.startswith('{"timing"'):
                t = json.loads(ln)['timing']
                print(tag, "hip %.3f eng %.3f first %.3f last %.3f written %.3f | setup %.3f waitf %.3f push_call %.3f wait_call %.3f first_push %.4f | H2D %.1f GB/s | written-hip %.3f -> %.1f Gbases/s" % (t['hip_ready'], t['engine_ready'], t['first_push'], t['last_push'], t['written'], t['stream_setup_s'], t['stream_wait_frame_s'], t['push_call_s'], t['wait_call_s'], t['first_push_call_s'], 8.0 / (t['last_push'] - t['first_push']), t['written'] - t['hip_ready'], 7.5 / (t['written'] - t['hip_ready'])), flush=True)
run('warm          ', {}, ['-p', '16'], reps=1)
run('p16 c8 if3    ', {}, ['-p', '16'])
run('p16 c8 if8    ', {}, ['-p', '16', '--inflight', '8'])
run('p16 c16 if8   ', {}, ['-p', '16', '--chunk-mib', '16', '--inflight', '8'])
run('p16 c32 if6   ', {}, ['-p', '16', '--chunk-mib', '32', '--inflight', '6'])
PY
cat gpurun_out/g_push_timing.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d gpurun_out/g_copytrace -- metakssd_amd/bin/metakssd dist -L /dev/shm/L3K11.shuf -A -o /dev/shm/o --quiet --timing -p 16 /dev/shm/big.fq > gpurun_out/g_copytrace.log 2>&1
tail -2 gpurun_out/g_copytrace.log
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/g_copytrace/**/*memory_copy_trace.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
print(rows[0].keys())
h2d = [r for r in rows if 'HOST_TO_DEVICE' in r.get('Direction', '') ]
print(len(rows), 'copies', len(h2d), 'h2d')
st = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in h2d)
big = [(a, b) for a, b in st if b - a > 20000]
dur = sum(b - a for a, b in big)
span = big[-1][1] - big[0][0]
gaps = sum(max(0, big[i + 1][0] - big[i][1]) for i in range(len(big) - 1))
print('big copies %d, busy %.3f s, span %.3f s, gaps %.3f s, rate while busy %.1f GB/s (assuming 8 GB)' % (len(big), dur / 1e9, span / 1e9, gaps / 1e9, 8.0 / (dur / 1e9)))
import statistics
print('median copy us', statistics.median([(b - a) / 1e3 for a, b in big]), 'p90', sorted([(b - a) / 1e3 for a, b in big])[int(len(big) * 0.9)])
PY
rm -rf /dev/shm/L3K11.shuf /dev/shm/big.fq /dev/shm/o
ls gpurun_out/g_copytrace/*/ | head
