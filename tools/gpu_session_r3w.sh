#!/bin/bash
# where the command line's start-up goes on a genome directory: engine-creation ticks of the tuning build (LD_PRELOAD) + --timing
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from metakssd_amd import capi
d = '/dev/shm/c5w'; os.makedirs(d + '/genomes', exist_ok=True)
rs = np.random.RandomState(5); acgt = np.frombuffer(b"ACGT", np.uint8)
for i in range(256):
    seq = acgt[rs.randint(0, 4, size=4000000)]
    with open('%s/genomes/g%03d.fna' % (d, i), 'wb') as f:
        f.write(b">g%d\n" % i); f.write(seq.tobytes()); f.write(b"\n")
capi.Shuf.generate(11, 5, 2, 211).write(d + '/L2K11.shuf')
capi.Shuf.generate(10, 6, 3, 10).write(d + '/L3K10.shuf')
PY
mkdir -p /tmp/pre && cp metakssd_amd/lib_tuning/tuning_plain.so /tmp/pre/libmetakssd_hip.so
for g in L3K10 L2K11; do
  for rep in 1 2 3; do
    echo "== $g rep $rep (shipped library)"
    metakssd_amd/bin/metakssd dist -L /dev/shm/c5w/$g.shuf -p 32 -o /dev/shm/c5w/out_$g --quiet --timing /dev/shm/c5w/genomes 2>&1 | tail -2 | cut -c1-700
    sleep 2
  done
  echo "== $g (tuning build preloaded: ticks)"
  LD_PRELOAD=/tmp/pre/libmetakssd_hip.so MK_DEBUG=1 metakssd_amd/bin/metakssd dist -L /dev/shm/c5w/$g.shuf -p 32 -o /dev/shm/c5w/out_$g --quiet --timing /dev/shm/c5w/genomes 2>&1 | tail -30 | cut -c1-700
  sleep 2
done
rm -rf /dev/shm/c5w
