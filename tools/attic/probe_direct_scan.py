#!/usr/bin/env python3
"""tools/probe_direct_scan.py -- does the scan kernel keep PCIe busy when it reads the rows straight out of pinned host memory
(no staging copy)?  The same 50 M reads in one pinned buffer, pushed as device pointers in pieces of several sizes."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from metakssd_amd import capi  # noqa: E402

N, STRIDE = int(os.environ.get("N_READS", "50000000")), 160
shuf = capi.Shuf.generate(11, 6, 3, 11)
eng = capi.Engine(shuf, 0)
dev = torch.device("cuda", 0)
reads = torch.empty(N * STRIDE, dtype=torch.uint8, device=dev)
capi.synth_rows_device(0, torch.cuda.current_stream().cuda_stream, 20261002, 0, N, 150, STRIDE, reads.data_ptr())
pinned = torch.empty(N * STRIDE, dtype=torch.uint8, pin_memory=True)
pinned.copy_(reads)
torch.cuda.synchronize()
out = {}
for piece in (27000, 54000, 108000, 432000, 1700000, N):
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        eng.begin(capi.MK_MODE_KOC)
        for lo in range(0, N, piece):
            n = min(piece, N - lo)
            eng.push_reads_device(pinned.data_ptr() + lo * STRIDE, STRIDE, n, lo)
        r = eng.finish_raw()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    out["direct_%d" % piece] = {"s": round(best, 4), "GBps": round(N * STRIDE / best / 1e9, 1), "distinct": int(r.total)}
    print(piece, out["direct_%d" % piece], flush=True)
# the staged path for comparison (hipMemcpyAsync into regions)
best = None
for rep in range(3):
    t0 = time.perf_counter()
    eng.begin(capi.MK_MODE_KOC)
    capi._check(capi.lib.mk_sketch_push_reads(eng.h, pinned.data_ptr(), STRIDE, N, 0), eng.h)
    r = eng.finish_raw()
    dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
out["staged_one_call"] = {"s": round(best, 4), "GBps": round(N * STRIDE / best / 1e9, 1), "distinct": int(r.total)}
print(json.dumps(out))
