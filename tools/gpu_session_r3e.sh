#!/bin/bash
# round 3, session e: bench line with the config 5 leg; kernel statistics of a genome-directory run (L3K10, L2K11)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python bench.py --steps 100 --warmup 5 > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err
tail -c 3000 gpurun_out/r3e_bench.json
# genomes for the profile: 128 x 4 Mbases
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from metakssd_amd import capi
d = "/dev/shm/mkprof5"; gd = d + "/genomes"
os.makedirs(gd, exist_ok=True)
rs = np.random.RandomState(5)
nl = int(3 * 4e6 / 70)
pool = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=(nl, 70))]
pool = np.concatenate([pool, np.full((nl, 1), 10, np.uint8)], axis=1).reshape(-1)
half = int(4e6 / 2 / 70)
for i in range(512):
    a = (i * 7919) % (nl - 2 * half - 1); b = (a + half + 1 + (i * 104729) % (nl - 2 * half - 1)) % (nl - half)
    with open("%s/g%04d.fna" % (gd, i), "wb") as f:
        f.write(b">g%d_0\n" % i); f.write(pool[71 * a: 71 * (a + half)].tobytes()); f.write(b">g%d_1\n" % i); f.write(pool[71 * b: 71 * (b + half)].tobytes())
capi.Shuf.generate(10, 6, 3, 10).write(d + "/L3K10.shuf")
capi.Shuf.generate(11, 5, 2, 211).write(d + "/L2K11.shuf")
PY
# engines taking the files in turn on GPU 0 (one driving thread), and the file-sharded drivers: 512 genomes, wall seconds
for S in L3K10 L2K11; do
  for V in "--engines 1" "--engines 2" "--devices 0,0"; do
    for rep in 1 2 3; do
      s=$(date +%s.%N)
      $GRAFT_REPO_ROOT/metakssd_amd/bin/metakssd dist -L /dev/shm/mkprof5/$S.shuf -p 16 $V -o /dev/shm/mkprof5/o_${S}_${rep} --quiet /dev/shm/mkprof5/genomes > /dev/null 2>&1
      e=$(date +%s.%N)
      echo "[$V] $S rep $rep: $(python3 -c "print(round($e-$s,3), 's ->', round(512/($e-$s)), 'genomes/s')")" | tee -a $GRAFT_REPO_ROOT/gpurun_out/r3e_engines.txt
      rm -rf /dev/shm/mkprof5/o_${S}_${rep}
    done
  done
done
cd /tmp && export TMPDIR=/tmp
for S in L3K10 L2K11; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3e_prof_$S
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3e_prof_$S -- $GRAFT_REPO_ROOT/metakssd_amd/bin/metakssd dist -L /dev/shm/mkprof5/$S.shuf -p 16 -o /dev/shm/mkprof5/out_$S --quiet --timing --slow-exit /dev/shm/mkprof5/genomes > $GRAFT_REPO_ROOT/gpurun_out/r3e_prof_$S.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3e_prof_$S -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r3e_config5_${S}_kernel_stats.csv && head -25 $f
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/r3e_prof_$S.log
done
rm -rf /dev/shm/mkprof5
