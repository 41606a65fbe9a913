#!/usr/bin/env python3
"""tools/probe_ragged_reads.py -- what real reads cost the scan kernel: the bench's 50 M rows of 150 bases (pitch 160) as they
are, trimmed to random lengths, and with an N in a fraction of the reads.  The tuned loop runs while a wave's 64 reads look
alike (same run length, no invalid base); everything else takes the wave-uniform scalar path or the predicated path."""
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
g = torch.Generator(device=dev); g.manual_seed(5)
col = torch.arange(160, device=dev, dtype=torch.int16)[None, :]


def variant(kind):
    rows = base.clone().view(N, 160)
    if kind.startswith("trim"):
        lo = int(kind[4:])
        for a in range(0, N, 5_000_000):  # in slices: the masks are 160 bytes per row
            b = min(N, a + 5_000_000)
            L = torch.randint(lo, 151, (b - a, 1), device=dev, generator=g, dtype=torch.int16)
            rows[a:b][col >= L] = 10
    elif kind.startswith("N"):
        frac = float(kind[1:])
        idx = torch.nonzero(torch.rand(N, device=dev, generator=g) < frac).flatten()
        pos = torch.randint(0, 150, (idx.numel(),), device=dev, generator=g)
        rows[idx, pos] = 78
    return rows.view(-1)


out = {}
for kind in ("asis", "trim149", "trim140", "trim100", "N0.001", "N0.01", "N0.05"):
    rows = base if kind == "asis" else variant(kind)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    for rep in range(6):
        if rep == 2:
            eng.profile_reset()
        eng.begin(capi.MK_MODE_KOC)
        eng.push_reads_device(rows.data_ptr(), 160, N, 0)
        r = eng.finish_raw()
        total = int(r.total)
        capi.lib.mk_result_release(eng.h, r)
    p = eng.profile()
    out[kind] = {"scan_ms": round(p["scan_ms"] / 4, 3), "resolve_ms": round(p["resolve_ms"] / 4, 3), "distinct": total}
    print(kind, out[kind], flush=True)
    if kind != "asis":
        del rows
print(json.dumps(out))
