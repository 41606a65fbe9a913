#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > gpurun_out/c_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c_pytest.log
tail -3 gpurun_out/c_pytest.log
./tools/probe_exit > gpurun_out/c_probe_exit.json 2>&1; cat gpurun_out/c_probe_exit.json
timeout 1200 python tools/bench_e2e.py --reads 50000000 --threads 12,16,24,32 --chunks 4,8 --reps 2 --out gpurun_out/c_e2e_sweep.json > gpurun_out/c_e2e_sweep.log 2>&1
grep SUMMARY gpurun_out/c_e2e_sweep.log
# engine-create breakdown from the tuning build
python - <<'PY' > gpurun_out/c_create_ticks.log 2>&1
import os, subprocess, sys
sys.path.insert(0, '.')
from metakssd_amd import capi
capi.Shuf.generate(11, 6, 3, 11).write('/dev/shm/L3K11.shuf')
capi.lib.mk_synth_fastq_write_mt(b'/dev/shm/small.fq', 1, 0, 2000000, 150, 16)
env = dict(os.environ, LD_LIBRARY_PATH=os.path.abspath('metakssd_amd/lib_tuning'), MK_DEBUG='1')
for i in range(2):
    r = subprocess.run(['metakssd_amd/bin/metakssd', 'dist', '-L', '/dev/shm/L3K11.shuf', '-A', '-o', '/dev/shm/o', '--quiet', '--timing', '-p', '16', '/dev/shm/small.fq'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    print(r.stdout.decode())
PY
cat gpurun_out/c_create_ticks.log | tail -12
rm -rf /dev/shm/L3K11.shuf /dev/shm/small.fq /dev/shm/o
