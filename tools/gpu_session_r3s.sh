#!/bin/bash
# where engine creation spends its time (tuning build: MK_DEBUG ticks), L3K10 and L2K11
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time
os.environ["MK_LIBRARY"] = os.path.join(os.environ["GRAFT_REPO_ROOT"], "metakssd_amd/lib_tuning/tuning_plain.so")
os.environ["MK_DEBUG"] = "1"
sys.path.insert(0, ".")
from metakssd_amd import capi
for name, (k, s, l, seed) in (("L3K10", (10, 6, 3, 10)), ("L2K11", (11, 5, 2, 211)), ("L3K11", (11, 6, 3, 11))):
    sh = capi.Shuf.generate(k, s, l, seed)
    for rep in range(2):
        t0 = time.time()
        e = capi.Engine(sh, 0)
        print("== %s rep %d: create %.1f ms" % (name, rep, 1e3 * (time.time() - t0)), flush=True)
        e.close()
PY
