#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY'
import sys
sys.path.insert(0, '.')
from metakssd_amd import capi
capi.Shuf.generate(11, 6, 3, 11).write('/dev/shm/L3K11.shuf')
capi.lib.mk_synth_fastq_write_mt(b'/dev/shm/big.fq', 20261002, 0, 50000000, 150, 64)
PY
metakssd_amd/bin/metakssd dist -L /dev/shm/L3K11.shuf -A -o /dev/shm/o --quiet -p 16 /dev/shm/big.fq
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d gpurun_out/h_copytrace -- metakssd_amd/bin/metakssd dist -L /dev/shm/L3K11.shuf -A -o /dev/shm/o --quiet --timing --slow-exit -p 16 /dev/shm/big.fq > gpurun_out/h_copytrace.log 2>&1
tail -1 gpurun_out/h_copytrace.log
python - <<'PY'
import csv, glob, statistics
f = glob.glob('gpurun_out/h_copytrace/**/*memory_copy_trace.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
print(list(rows[0].keys()))
print(rows[5])
st = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
big = [(a, b) for a, b in st if b - a > 20000]
dur = sum(b - a for a, b in big)
span = big[-1][1] - big[0][0]
gaps = sum(max(0, big[i + 1][0] - big[i][1]) for i in range(len(big) - 1))
print('copies %d big %d, busy %.3f s, span %.3f s, gaps %.3f s, rate while busy %.1f GB/s (8 GB)' % (len(rows), len(big), dur / 1e9, span / 1e9, gaps / 1e9, 8.0 / (dur / 1e9)))
d = sorted((b - a) / 1e3 for a, b in big)
print('copy us: min %.0f median %.0f p90 %.0f max %.0f' % (d[0], statistics.median(d), d[int(len(d) * 0.9)], d[-1]))
g = sorted(max(0, big[i + 1][0] - big[i][1]) / 1e3 for i in range(len(big) - 1))
print('gap us: median %.1f p90 %.1f p99 %.1f max %.1f' % (statistics.median(g), g[int(len(g) * 0.9)], g[int(len(g) * 0.99)], g[-1]))
# first 12 copies timeline
t0 = big[0][0]
print([(round((a - t0) / 1e3), round((b - a) / 1e3)) for a, b in big[:12]])
k = glob.glob('gpurun_out/h_copytrace/**/*kernel_trace.csv', recursive=True)
kr = list(csv.DictReader(open(k[0])))
sc = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in kr if 'mk_scan' in r['Kernel_Name']]
print('scan launches', len(sc), 'median us', statistics.median([(b - a) / 1e3 for a, b in sc]))
PY
rm -rf /dev/shm/L3K11.shuf /dev/shm/big.fq /dev/shm/o gpurun_out/h_copytrace
