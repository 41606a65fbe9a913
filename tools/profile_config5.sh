#!/bin/bash
# kernel statistics of a genome-directory run (config 5: 256 genomes of 4 Mbases) at L2K11 and L3K10 -> gpurun_out/c5_<geometry>_kernel_stats.csv
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from metakssd_amd import capi
d = '/dev/shm/c5p'; os.makedirs(d + '/genomes', exist_ok=True)
rs = np.random.RandomState(5); acgt = np.frombuffer(b"ACGT", np.uint8)
for i in range(256):
    n = 4000000
    seq = acgt[rs.randint(0, 4, size=n)]
    with open('%s/genomes/g%03d.fna' % (d, i), 'wb') as f:
        f.write(b">g%d\n" % i); f.write(seq.tobytes()); f.write(b"\n")
capi.Shuf.generate(11, 5, 2, 211).write(d + '/L2K11.shuf')
capi.Shuf.generate(10, 6, 3, 10).write(d + '/L3K10.shuf')
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in L2K11 L3K10; do
rm -rf gpurun_out/c5_prof_$g
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5_prof_$g -- metakssd_amd/bin/metakssd dist -L /dev/shm/c5p/$g.shuf -p 32 -o /dev/shm/c5p/out_$g --quiet --slow-exit /dev/shm/c5p/genomes > gpurun_out/c5_$g.log 2>&1
# the NEWEST statistics file (a directory left by an earlier round holds older pids)
f=$(ls -t $(find gpurun_out/c5_prof_$g -name "*kernel_stats.csv") | head -1); echo "== $g: $f"; head -16 $f | cut -c1-150
cp $f gpurun_out/c5_${g}_kernel_stats.csv
done
rm -rf /dev/shm/c5p
