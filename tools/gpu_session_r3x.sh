#!/bin/bash
# config 5 with 1..4 engines in turn, by the command line's own clock (--timing): written - engine_ready over 1024 genomes
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, json, re, subprocess, time, numpy as np
sys.path.insert(0, '.')
from metakssd_amd import capi
d = '/dev/shm/c5w'; os.makedirs(d + '/genomes', exist_ok=True)
rs = np.random.RandomState(5); acgt = np.frombuffer(b"ACGT", np.uint8)
for i in range(1024):
    seq = acgt[rs.randint(0, 4, size=4000000)]
    with open('%s/genomes/g%04d.fna' % (d, i), 'wb') as f:
        f.write(b">g%d\n" % i); f.write(seq.tobytes()); f.write(b"\n")
capi.Shuf.generate(11, 5, 2, 211).write(d + '/L2K11.shuf')
capi.Shuf.generate(10, 6, 3, 10).write(d + '/L3K10.shuf')
VARIANTS = [["--engines", "2"], ["--devices", "0,0"], ["--devices", "0,0,0"], ["--devices", "0,0,0,0"]] * 3
for g in ("L3K10", "L2K11"):
    for n in VARIANTS:
        t0 = time.perf_counter()
        p = subprocess.run(["metakssd_amd/bin/metakssd", "dist", "-L", "%s/%s.shuf" % (d, g), "-p", "16", *n, "-o", "%s/out_%s" % (d, g),
                            "--quiet", "--timing", d + "/genomes"], capture_output=True, text=True, stdin=subprocess.DEVNULL)
        wall = time.perf_counter() - t0
        m = re.search(r'\{"timing".*\}', p.stdout + p.stderr)
        if not m:
            print(g, n, "no timing line; rc", p.returncode, (p.stdout + p.stderr)[-300:]); continue
        t = json.loads(m.group(0))["timing"]
        print(g, " ".join(n), "engine_ready %.3f written %.3f -> %.0f genomes/s after engine ready, %.0f by written, %.0f by wall (%.3f s)" %
              (t["engine_ready"], t["written"], 1024 / (t["written"] - t["engine_ready"]), 1024 / t["written"], 1024 / wall, wall), flush=True)
        time.sleep(3)
PY
rm -rf /dev/shm/c5w
