#!/usr/bin/env python3
"""NPARTS engines in ONE process, each sketching its contiguous range; exports folded into engine 0; compared with a
single-engine sketch of everything.  Repeated REPS times to expose timing-dependent behaviour."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from metakssd_amd import capi
N = int(os.environ.get("N_READS", "400000000")); P = int(os.environ.get("NPARTS", "8")); REPS = int(os.environ.get("REPS", "2"))
dev = torch.device("cuda", 0)
shuf = capi.Shuf.generate(11, 6, 3, 11)
reads = torch.empty(N * 160, dtype=torch.uint8, device=dev)
capi.synth_rows_device(0, None, 20261002, 0, N, 150, 160, reads.data_ptr()); torch.cuda.synchronize()
def same(a, b): return all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))
engs = [capi.Engine(shuf, 0) for _ in range(P)]
e = engs[0]
e.begin(); e.push_reads_device(reads.data_ptr(), 160, N, 0); A = e.finish()
print("single distinct", len(A[0][0]))
per = N // P
for rep in range(REPS):
    for r, g in enumerate(engs):
        g.begin(); g.push_reads_device(reads.data_ptr() + r * per * 160, 160, per if r < P - 1 else N - per * (P - 1), r * per)
    for g in engs[1:]:
        d = g.partial_count()
        k = torch.empty(d, dtype=torch.int64, device=dev); c = torch.empty(d, dtype=torch.int32, device=dev); o = torch.empty(d, dtype=torch.int64, device=dev)
        g.partial_export(k.data_ptr(), c.data_ptr(), o.data_ptr(), d)
        e.partial_import(k.data_ptr(), c.data_ptr(), o.data_ptr(), d); torch.cuda.synchronize()
    M = e.finish()
    e.begin(); e.push_reads_device(reads.data_ptr(), 160, N, 0); A2 = e.finish()
    print("rep", rep, "merged distinct", len(M[0][0]), "merged == single:", same(M, A), " single again == single:", same(A2, A))
