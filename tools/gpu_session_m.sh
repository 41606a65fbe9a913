#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_golden.py -m gpu -x -q -k "several_engines or dropin" > gpurun_out/m_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/m_pytest.log
tail -3 gpurun_out/m_pytest.log
# BASELINE config 4's table regime on one GPU: 500 M reads (80 GB of rows), whole vs 8 shards merged through export/import
timeout 1200 python bench.py --total-reads 500000000 --steps 5 --warmup 1 --verify --no-cpu-baseline --no-host-legs > gpurun_out/m_config4_one_gpu.json 2> gpurun_out/m_config4.err; echo "config4 rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/m_config4_one_gpu.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'scaling')}, d['config']['distinct_keys'], d['config']['table_load'], d.get('merged_equals_single_engine'), d['phases_ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/m_prof -- python3 bench.py --total-reads 500000000 --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs > gpurun_out/m_prof.log 2>&1
find gpurun_out/m_prof -name "*kernel_stats.csv" | head -1 | xargs head -9
# the C product over several engines on this one GPU, 50 M-read file
python - <<'PY' > gpurun_out/m_multi_cli.log 2>&1
import os, subprocess, sys, json, time, hashlib
sys.path.insert(0, '.')
from metakssd_amd import capi
capi.Shuf.generate(11, 6, 3, 11).write('/dev/shm/L3K11.shuf')
capi.lib.mk_synth_fastq_write_mt(b'/dev/shm/big.fq', 20261002, 0, 50000000, 150, 64)
ref = None
for devs in (None, '0,0', '0,0,0,0', '0,0,0,0,0,0,0,0'):
    for rep in range(2):
        time.sleep(1.0)
        cmd = ['metakssd_amd/bin/metakssd', 'dist', '-L', '/dev/shm/L3K11.shuf', '-A', '-o', '/dev/shm/o', '--quiet', '--timing'] + (['--devices', devs] if devs else []) + ['/dev/shm/big.fq']
        t0 = time.perf_counter(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT); wall = time.perf_counter() - t0
        h = hashlib.sha256(open('/dev/shm/o/combco.0','rb').read() + open('/dev/shm/o/combco.0.a','rb').read()).hexdigest()[:12]
        ref = ref or h
        t = [json.loads(ln)['timing'] for ln in r.stdout.decode().splitlines() if ln.startswith('{"timing"')]
        t = t[0] if t else {}
        print(json.dumps({'devices': devs or 'single', 'rc': r.returncode, 'same_sketch': h == ref, 'wall_s': round(wall, 3), 'written_minus_init_s': round(t.get('written', 0) - t.get('hip_ready', 0), 4), 'engine_ready': t.get('engine_ready'), 'gather_ms': t.get('gather_ms'), 'tail_ms': t.get('tail_ms'), 'transport': t.get('transport'), 'err': r.stdout.decode()[-200:] if r.returncode else ''}), flush=True)
PY
cat gpurun_out/m_multi_cli.log
rm -rf /dev/shm/L3K11.shuf /dev/shm/big.fq /dev/shm/o
