#!/usr/bin/env python3
"""tools/bench_set.py -- `set -u` / `set -q` on the device: time per union of N ids resident in HBM.

    python tools/bench_set.py [--ids 200000000] [--space-bits 32] [--steps 5] [--cpu-sample 20000000]

Prints one JSON line per mode: M ids/s, the phase times, the HBM-roofline fraction of the dictionary walk
(count pass streams 2^29 bytes, 2^30 for -q) and, on a bounded sample, the oracle's CPU restatement of the
reference loop (command_set.c:293-312) beside it."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ids", type=int, default=200_000_000)
    ap.add_argument("--space-bits", type=int, default=32, help="ids are uniform in [0, 2^bits)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=20_000_000)
    args = ap.parse_args()
    import torch
    from metakssd_amd import capi
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(20261003)
    ids = torch.randint(0, 2 ** args.space_bits, (args.ids,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    torch.cuda.synchronize()
    so = capi.SetOp(0)
    for uniq in (False, True):
        times = []
        n_out = 0
        for it in range(args.steps + 1):
            t0 = time.perf_counter()
            so.begin(uniq=uniq)
            so.add_device(ids.data_ptr(), ids.numel())
            n_out = so.finish_count()
            t1 = time.perf_counter()
            if it:
                times.append(t1 - t0)
        ms = 1e3 * sum(times) / len(times)
        line = {"op": "set -q" if uniq else "set -u", "ids": args.ids, "space_bits": args.space_bits, "out_ids": n_out,
                "ms_per_union": ms, "value": args.ids / (ms * 1e-3) / 1e6, "unit": "M ids/s",
                "note": "begin (dictionary clear) + mark + count + scan + write + result copy to the host; input resident in HBM"}
        if args.cpu_sample:
            from oracle_binding import load
            lib = load()
            lib.ko_set_union.restype = C.c_size_t
            lib.ko_set_union.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
            sample = ids[: args.cpu_sample].cpu().numpy().view(np.uint32)
            out = np.zeros(sample.size, np.uint32)
            t0 = time.perf_counter()
            m = lib.ko_set_union(sample.ctypes.data, sample.size, 1 if uniq else 0, out.ctypes.data)
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": sample.size / dt / 1e6, "unit": "M ids/s", "cores": 1, "kind": "port",
                                    "sample": "first %d ids, oracle restatement of command_set.c:279-316 (%.2f s)" % (sample.size, dt),
                                    "out_ids": int(m)}
        print(json.dumps(line))
    so.close()


if __name__ == "__main__":
    main()
