cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from metakssd_amd import capi
d = '/dev/shm/c5q'; os.makedirs(d + '/genomes', exist_ok=True)
rs = np.random.RandomState(5); acgt = np.frombuffer(b"ACGT", np.uint8)
for i in range(64):
    seq = acgt[rs.randint(0, 4, size=4000000)]
    with open('%s/genomes/g%03d.fna' % (d, i), 'wb') as f:
        f.write(b">g%d\n" % i); f.write(seq.tobytes()); f.write(b"\n")
capi.Shuf.generate(11, 5, 2, 211).write(d + '/L2K11.shuf')
capi.Shuf.generate(10, 6, 3, 10).write(d + '/L3K10.shuf')
PY
for g in L3K10 L2K11 L3K10 L2K11; do
  echo "== $g"
  LD_LIBRARY_PATH=$PWD/metakssd_amd/lib_tuning/base MK_DEBUG=1 metakssd_amd/bin/metakssd dist -L /dev/shm/c5q/$g.shuf -o /dev/shm/c5q/out_$g --quiet --timing /dev/shm/c5q/genomes 2>&1 | grep -E "engine create|timing" | cut -c1-400
  sleep 2
done
rm -rf /dev/shm/c5q
