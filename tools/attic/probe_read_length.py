#!/usr/bin/env python3
"""tools/probe_read_length.py -- scan + resolve time per base at other read lengths than the bench's 150 (rows resident in HBM,
stride = read length + 1 rounded up to 16, as the FASTQ front end lays them out).  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from metakssd_amd import capi  # noqa: E402

BASES = float(os.environ.get("GBASES", "7.5")) * 1e9
shuf = capi.Shuf.generate(11, 6, 3, 11)
eng = capi.Engine(shuf, 0)
dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
out = {}
LENS = [int(x) for x in os.environ.get("LENS", "75,100,125,150,200,250,300").split(",")]
REPS = int(os.environ.get("REPS", "8"))
PAD = int(os.environ.get("PAD", "0"))  # extra 16-byte units of row pitch
for L in LENS:
    stride = (L + 1 + 15) // 16 * 16 + 16 * PAD
    n = int(BASES // L)
    rows = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    capi.synth_rows_device(0, torch.cuda.current_stream().cuda_stream, 20261002, 0, n, L, stride, rows.data_ptr())
    torch.cuda.synchronize()
    eng.profile_enable(True)
    for rep in range(REPS):
        if rep == 2:
            eng.profile_reset()
        eng.begin(capi.MK_MODE_KOC)
        eng.push_reads_device(rows.data_ptr(), stride, n, 0)
        r = eng.finish_raw()
        total = int(r.total)
        capi.lib.mk_result_release(eng.h, r)
    p = eng.profile()
    k = max(1, REPS - 2)  # sketches timed (a push above 67 M reads is several scan launches: totals per sketch, not per launch)
    scan_ms = p["scan_ms"] / k
    out["len_%d_stride_%d" % (L, stride)] = {"stride": stride, "reads": n, "launches_per_sketch": p["scan_launches"] // k, "scan_ms": round(scan_ms, 3), "resolve_ms": round(p["resolve_ms"] / k, 3),
                         "scan_gbases_s": round(n * L / scan_ms / 1e6, 1), "hbm_gb_s": round(n * stride / scan_ms / 1e6, 1), "distinct": total}
    print(L, out["len_%d_stride_%d" % (L, stride)], flush=True)
    del rows
print(json.dumps(out))
