#!/usr/bin/env python3
"""tools/probe_row_stride.py -- the same 50 M reads of 150 bases as rows of 160 bytes (16-byte staging, the bench layout),
152 bytes (no padding beyond the newline and one byte: 4-byte staging) and 176 / 192 bytes: what does the row pitch cost
the scan kernel?  Prints one JSON line: per stride the HIP-event time of the scan and the sketch's size."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from metakssd_amd import capi  # noqa: E402

N = int(os.environ.get("N_READS", "50000000"))
shuf = capi.Shuf.generate(11, 6, 3, 11)
eng = capi.Engine(shuf, 0)
dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
base = torch.empty(N * 160, dtype=torch.uint8, device=dev)
capi.synth_rows_device(0, torch.cuda.current_stream().cuda_stream, 20261002, 0, N, 150, 160, base.data_ptr())
torch.cuda.synchronize()
out = {}
for stride in (160, 152, 176, 192):
    if stride == 160:
        rows = base
    elif stride < 160:
        rows = base.view(N, 160)[:, :stride].contiguous().view(-1)
    else:
        rows = torch.full((N, stride), 10, dtype=torch.uint8, device=dev)  # '\n' padding
        rows[:, :160] = base.view(N, 160)
        rows = rows.view(-1)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    for rep in range(12):
        if rep == 2:
            eng.profile_reset()
        eng.begin(capi.MK_MODE_KOC)
        eng.push_reads_device(rows.data_ptr(), stride, N, 0)
        r = eng.finish_raw()
        total = int(r.total)
        capi.lib.mk_result_release(eng.h, r)
    p = eng.profile()
    out["stride_%d" % stride] = {"scan_ms": round(p["scan_ms"] / max(1, p["scan_launches"]), 4),
                                 "resolve_ms": round(p["resolve_ms"] / max(1, p["scan_launches"]), 4), "distinct": total}
    print(stride, out["stride_%d" % stride], flush=True)
    if stride != 160:
        del rows
print(json.dumps(out))
