#!/usr/bin/env python3
"""tools/bench_stream.py -- the two host-inclusive rates next to bench.py's HBM-resident number:
  t_stream : rows in PINNED host memory -> mk_sketch_push_reads (H2D double-buffered) -> finish
  t_e2e    : `metakssd dist -L L3K11.shuf -A` on a FASTQ file in /dev/shm (parsing + H2D + sketch + files)
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metakssd_amd import capi  # noqa: E402

N = int(os.environ.get("N_READS", "20000000"))
STRIDE, LEN = 160, 150
shuf = capi.Shuf.generate(11, 6, 3, 11)
eng = capi.Engine(shuf, 0)
p = C.c_void_p()
assert capi.lib.mk_host_alloc(C.byref(p), N * STRIDE) == 0
t0 = time.perf_counter()
capi.lib.mk_synth_rows_host(1, 0, N, LEN, STRIDE, p)
print("host generator: %.2f s for %d reads" % (time.perf_counter() - t0, N))
for rep in range(3):
    t0 = time.perf_counter()
    eng.begin(capi.MK_MODE_KOC)
    capi._check(capi.lib.mk_sketch_push_reads(eng.h, p, STRIDE, N, 0), eng.h)
    r = eng.finish_raw()
    dt = time.perf_counter() - t0
    print("t_stream rep %d: %.3f s  %.1f Gbases/s  (%.1f GB/s H2D)  distinct %d" % (rep, dt, N * LEN / dt / 1e9, N * STRIDE / dt / 1e9, r.total))
capi.lib.mk_host_free(p)
eng.close()

d = tempfile.mkdtemp(prefix="mke2e_", dir="/dev/shm")
fq, sp = os.path.join(d, "in.fq"), os.path.join(d, "L3K11.shuf")
M = min(N, 10000000)
capi.lib.mk_synth_fastq_write(fq.encode(), 1, 0, M, LEN)
shuf.write(sp)
cli = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
for rep in range(2):
    t0 = time.perf_counter()
    subprocess.check_call([cli, "dist", "-L", sp, "-A", "-o", os.path.join(d, "out%d" % rep), "--quiet", fq])
    dt = time.perf_counter() - t0
    print("t_e2e rep %d: %.3f s for %d reads (%.2f GB FASTQ)  %.2f Gbases/s" % (rep, dt, M, os.path.getsize(fq) / 1e9, M * LEN / dt / 1e9))
subprocess.call(["rm", "-rf", d])
