#!/bin/bash
# one genome at a time (the reference permutes multi-file inputs): product CLI vs the compiled reference, byte for byte,
# on 4-Mbase synthetic genomes with the large L2K11 table (sparse bookkeeping) and L3K10
cd $GRAFT_REPO_ROOT
W=$(mktemp -d -p /dev/shm)
python3 - "$W" <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from metakssd_amd import capi
w = sys.argv[1]
rs = np.random.RandomState(3)
acgt = np.frombuffer(b"ACGT", np.uint8)
for i in range(3):
    n = 4_000_000 + 1000 * i
    s = acgt[rs.randint(0, 4, size=n)].tobytes()
    with open("%s/g%d.fna" % (w, i), "wb") as f:
        f.write(b">c0\n" + b"\n".join(s[j:j + 70] for j in range(0, n // 2, 70)) + b"\n>c1 x\n" + b"\n".join(s[j:j + 70] for j in range(n // 2, n, 70)) + b"\n")
capi.Shuf.generate(11, 5, 2, 211).write(w + "/L2K11.shuf")
capi.Shuf.generate(10, 6, 3, 10).write(w + "/L3K10.shuf")
PY
ok=1
for sh in L2K11 L3K10; do
  for i in 0 1 2; do
    for u in "" "-u"; do
      oracle/_ref/metakssd dist -L $W/$sh.shuf $u -p 1 -o $W/ref_${sh}_$i$u $W/g$i.fna > /dev/null 2>&1
      metakssd_amd/bin/metakssd dist -L $W/$sh.shuf $u -o $W/out_${sh}_$i$u --quiet $W/g$i.fna > /dev/null 2>&1
      for f in $(ls $W/ref_${sh}_$i$u | grep combco); do
        cmp -s $W/ref_${sh}_$i$u/$f $W/out_${sh}_$i$u/$f || { echo "DIFF $sh g$i $u $f"; ok=0; }
      done
    done
  done
  echo "$sh: $(ls $W/out_${sh}_0 | wc -l) files per sketch, ids in g0: $(( $(cat $W/out_${sh}_0/combco.[0-9]* | wc -c) / 4 ))"
done
# and all three genomes in one run against three single runs concatenated (engine reuse with sparse clears in between)
metakssd_amd/bin/metakssd dist -L $W/L2K11.shuf -o $W/out_all --quiet $W/g0.fna $W/g1.fna $W/g2.fna > /dev/null 2>&1
for c in 0 7 15; do cat $W/out_L2K11_0/combco.$c $W/out_L2K11_1/combco.$c $W/out_L2K11_2/combco.$c | cmp -s - $W/out_all/combco.$c || { echo "DIFF multi comp $c"; ok=0; }; done
[ $ok = 1 ] && echo "ALL IDENTICAL"
rm -rf $W
