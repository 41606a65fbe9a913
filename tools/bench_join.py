#!/usr/bin/env python3
"""tools/bench_join.py -- composite -q's join (mk_setop_join) at MarkerDB scale: R reference ids in B blocks against a
query sketch of Q ids with counts, host arrays in, counts + segment boundaries out.

    python tools/bench_join.py [--ref-ids 100000000] [--blocks 20000] [--query-ids 1573525] [--steps 5]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-ids", type=int, default=100_000_000)
    ap.add_argument("--blocks", type=int, default=20000)
    ap.add_argument("--query-ids", type=int, default=1573525)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    from metakssd_amd import capi
    lib = capi.lib
    lib.mk_setop_join.restype = C.c_int
    lib.mk_setop_join.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    rs = np.random.RandomState(5)
    q = np.unique(rs.randint(0, 2 ** 32, size=args.query_ids, dtype=np.uint64).astype(np.uint32))
    qab = rs.randint(1, 200, size=q.size).astype(np.uint16)
    ref = rs.randint(0, 2 ** 32, size=args.ref_ids, dtype=np.uint64).astype(np.uint32)
    ref[:: 50] = q[rs.randint(0, q.size, size=ref[::50].size)]            # 2 % of the reference ids are in the query
    bounds = np.sort(np.concatenate([[0, ref.size], rs.randint(0, ref.size, size=args.blocks - 1)])).astype(np.uint64)
    bout = np.zeros(bounds.size, np.uint64)
    so = capi.SetOp(0)
    times = []
    n = C.c_uint64(0)
    for it in range(args.steps + 1):
        out = C.c_void_p()
        t0 = time.perf_counter()
        rc = lib.mk_setop_join(so.h, q.ctypes.data, qab.ctypes.data, q.size, ref.ctypes.data, ref.size, bounds.ctypes.data, bounds.size,
                               C.byref(out), C.byref(n), bout.ctypes.data)
        t1 = time.perf_counter()
        assert rc == 0
        if it:
            times.append(t1 - t0)
    ms = 1e3 * sum(times) / len(times)
    print(json.dumps({"op": "composite join", "ref_ids": int(ref.size), "blocks": int(bounds.size - 1), "query_ids": int(q.size),
                      "matches": int(n.value), "ms_per_join": ms, "value": ref.size / (ms * 1e-3) / 1e6, "unit": "M reference ids/s",
                      "note": "host arrays in (pageable numpy memory: the H2D copy of the reference ids is inside), counts and segment boundaries out"}))
    so.close()


if __name__ == "__main__":
    main()
