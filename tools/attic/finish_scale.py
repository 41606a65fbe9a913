import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from metakssd_amd import capi
dev = torch.device("cuda", 0)
shuf = capi.Shuf.generate(11, 6, 3, 11)
for N in (50_000_000, 200_000_000, 400_000_000):
    reads = torch.empty(N * 160, dtype=torch.uint8, device=dev)
    capi.synth_rows_device(0, None, 20261002, 0, N, 150, 160, reads.data_ptr()); torch.cuda.synchronize()
    e = capi.Engine(shuf, 0); e.profile_enable(True)
    for rep in range(2):
        e.profile_reset(); e.begin(); e.push_reads_device(reads.data_ptr(), 160, N, 0); r = e.finish_raw(); p = e.profile()
    print(N, "distinct", r.total, {k: round(v, 3) for k, v in p.items() if k.endswith("_ms")})
    e.close(); del reads
