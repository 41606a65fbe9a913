#!/usr/bin/env python3
"""one-off consistency check at scale (single GPU): the same N reads sketched three ways must give identical bytes:
   (a) one device push, (b) CHUNKS device pushes, (c) two engines over halves + export/import"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from metakssd_amd import capi
N = int(os.environ.get("N_READS", "100000000")); CHUNKS = int(os.environ.get("CHUNKS", "4"))
dev = torch.device("cuda", 0)
shuf = capi.Shuf.generate(11, 6, 3, 11)
reads = torch.empty(N * 160, dtype=torch.uint8, device=dev)
capi.synth_rows_device(0, None, 20261002, 0, N, 150, 160, reads.data_ptr()); torch.cuda.synchronize()
def same(a, b): return all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))
e = capi.Engine(shuf, 0)
e.begin(); e.push_reads_device(reads.data_ptr(), 160, N, 0); A = e.finish()
e.begin()
per = N // CHUNKS
for c in range(CHUNKS):
    n = per if c < CHUNKS - 1 else N - per * (CHUNKS - 1)
    e.push_reads_device(reads.data_ptr() + c * per * 160, 160, n, c * per)
B = e.finish()
print("distinct", len(A[0][0]), "one push == %d pushes:" % CHUNKS, same(A, B))
e2 = capi.Engine(shuf, 0)
half = N // 2
e.begin(); e.push_reads_device(reads.data_ptr(), 160, half, 0)
e2.begin(); e2.push_reads_device(reads.data_ptr() + half * 160, 160, N - half, half)
d = e2.partial_count()
k = torch.empty(d, dtype=torch.int64, device=dev); c = torch.empty(d, dtype=torch.int32, device=dev); o = torch.empty(d, dtype=torch.int64, device=dev)
e2.partial_export(k.data_ptr(), c.data_ptr(), o.data_ptr(), d)
e.partial_import(k.data_ptr(), c.data_ptr(), o.data_ptr(), d); torch.cuda.synchronize()
Cm = e.finish()
print("merged == chunked:", same(Cm, B), " merged == one push:", same(Cm, A))
if not same(A, B):
    a, b = A[0], B[0]
    print("len", len(a[0]), len(b[0]), "sorted-id equal:", np.array_equal(np.sort(a[0]), np.sort(b[0])), "count sum", int(a[1].astype(np.int64).sum()), int(b[1].astype(np.int64).sum()))
