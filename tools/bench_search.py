#!/usr/bin/env python3
"""tools/bench_search.py -- stage II (mk_mco_build, dense index slabs) and the `dist -r` counting loop (mk_mco_count_*) at
reference-database scale: R genomes of about G ids drawn as windows of one universe (neighbouring genomes share most ids,
like strains of a species), Q query sketches taken from the genomes themselves.  Host arrays in, results out; the oracle's
restatement of the reference loops is timed on a bounded sample beside it.

    python tools/bench_search.py [--refs 20000] [--ids 20000] [--queries 2000] [--steps 3] [--cpu-sample 200]"""
import argparse
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
    ap.add_argument("--refs", type=int, default=20000)
    ap.add_argument("--ids", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=200, help="reference genomes / 10 queries for the CPU restatement (0: skip)")
    ap.add_argument("--slabs", type=int, default=4, help="dense-index slabs (1 GiB each) to time")
    args = ap.parse_args()
    from metakssd_amd import capi
    rs = np.random.RandomState(9)
    R, G, Q = args.refs, args.ids, args.queries
    universe = np.unique(rs.randint(0, 2 ** 32, size=R * 40 + 4 * G, dtype=np.uint64).astype(np.uint32))
    sizes = rs.randint(G // 2, G + G // 2, size=R)
    starts = np.sort(rs.randint(0, universe.size - 2 * G, size=R))
    parts = []
    for s, n in zip(starts, sizes):       # a window of the universe, in hash-table (= arbitrary) order
        parts.append(rs.permutation(universe[s:s + n]))
    ids = np.concatenate(parts)
    index = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    sel = rs.randint(0, R, size=Q)
    qids = np.concatenate([parts[i] for i in sel])
    qindex = np.concatenate([[0], np.cumsum(sizes[sel])]).astype(np.uint64)
    qctx = sizes[sel].astype(np.uint32)
    m = capi.Mco(0)
    out = {"op": "stage II + dist -r counting", "refs": R, "ref_ids": int(ids.size), "queries": Q, "query_ids": int(qids.size)}
    t = []
    for it in range(args.steps + 1):
        t0 = time.perf_counter()
        gids, ri, re_ = m.build(ids, index, copy=False)   # the library's pinned result buffers, as the C caller gets them
        t.append(time.perf_counter() - t0)
    out["build_ms"] = 1e3 * min(t[1:])
    out["build_M_ids_per_s"] = ids.size / min(t[1:]) / 1e6
    out["rows"] = int(ri.size)
    t = []
    for it in range(args.slabs + 1):
        t0 = time.perf_counter()
        m.index_rows(it << 27, 1 << 27, pinned=True)    # as the command line: a pinned slab buffer
        t.append(time.perf_counter() - t0)
    out["index_slab_ms"] = 1e3 * min(t[1:])
    out["index_GBps_to_host"] = (8 << 27) / min(t[1:]) / 1e9
    t = []
    for it in range(2):
        t0 = time.perf_counter()
        m.index_rows(it << 27, 1 << 27)                 # into ordinary (pageable) memory
        t.append(time.perf_counter() - t0)
    out["index_slab_ms_pageable"] = 1e3 * min(t)
    t = []
    for it in range(args.steps + 1):
        t0 = time.perf_counter()
        ct = m.count(R, qindex, qctx, [{"qry_ids": qids}])
        t.append(time.perf_counter() - t0)
    incr = int(ct.sum(dtype=np.uint64))
    out["count_ms"] = 1e3 * min(t[1:])
    out["increments"] = incr
    out["count_G_increments_per_s"] = incr / min(t[1:]) / 1e9
    assert all(ct[k, sel[k]] == sizes[sel[k]] for k in range(0, Q, max(1, Q // 50)))   # a genome shares all its ids with itself
    if args.cpu_sample:
        import oracle_binding as ob
        rc = min(args.cpu_sample, R)
        sub = ids[:int(index[rc])]
        t0 = time.perf_counter()
        og, ori, ore = ob.mco_build(sub, index[:rc + 1])
        tb = time.perf_counter() - t0
        qn = max(1, min(Q, args.cpu_sample // 10))
        near = [k for k in range(Q) if sel[k] < rc][:qn] or [0]
        sq = np.concatenate([parts[sel[k]] for k in near])
        sqi = np.concatenate([[0], np.cumsum([parts[sel[k]].size for k in near])]).astype(np.uint64)
        t0 = time.perf_counter()
        oct_ = ob.mco_count(og, ori, ore, sq, sqi, np.diff(sqi).astype(np.uint32), rc)
        tc = time.perf_counter() - t0
        out["cpu"] = {"kind": "port (oracle restatement, 1 core)", "build_M_ids_per_s": sub.size / tb / 1e6,
                      "count_G_increments_per_s": int(oct_.sum(dtype=np.uint64)) / max(tc, 1e-9) / 1e9,
                      "sample": "%d reference genomes (%d ids), %d query sketches" % (rc, sub.size, len(near))}
    out["note"] = ("host arrays in (pageable numpy memory), results in (the library's pinned) host memory: copies are inside every number; "
                   "build = gid kernel + radix sort + row table + 3 result copies; count = extents on the device + counting kernel + matrix copy")
    print(json.dumps(out))
    m.close()


if __name__ == "__main__":
    main()
