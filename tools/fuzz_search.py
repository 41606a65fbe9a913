#!/usr/bin/env python3
"""tools/fuzz_search.py -- randomized stage II / search parity: mk_mco_build, mk_mco_index_rows, mk_mco_count_* (device) against
the oracle's ko_mco_build / ko_mco_count on random databases: empty and tiny sketches, repeated ids inside a sketch, ids at
both ends of the 32-bit range, 1 .. 70 000 genomes, both counter paths, extents given or looked up on the device.

    python tools/fuzz_search.py [--seconds 120] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def draw(rs, pool, nsk, lo, hi, p_empty, p_dup):
    parts, index = [], [0]
    for _ in range(nsk):
        n = 0 if rs.rand() < p_empty else int(rs.randint(lo, hi + 1))
        p = pool[rs.randint(0, pool.size, size=n)] if n else np.zeros(0, np.uint32)
        if rs.rand() >= p_dup:
            p = np.unique(p)
            p = p[rs.permutation(p.size)]
        parts.append(p.astype(np.uint32))
        index.append(index[-1] + p.size)
    return (np.concatenate(parts) if parts else np.zeros(0, np.uint32)).astype(np.uint32), np.array(index, np.uint64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import oracle_binding as ob
    from metakssd_amd import capi
    m = capi.Mco(0)
    rs = np.random.RandomState(args.seed)
    t0, it, incr = time.time(), 0, 0
    while time.time() - t0 < args.seconds:
        it += 1
        shape = rs.randint(0, 5)
        nref = int([rs.randint(1, 8), rs.randint(8, 300), rs.randint(300, 3000), rs.randint(33000, 70001), rs.randint(1, 50)][shape])
        hi = int([3000, 800, 120, 6, 20000][shape])
        universe = int(rs.randint(max(8, hi // 4), max(16, hi * [2, 6, 12, 4000, 3][shape])))
        pool = np.unique(rs.randint(0, 2 ** 32, size=universe, dtype=np.uint64).astype(np.uint32))
        if rs.rand() < 0.3:
            pool = np.unique(np.concatenate([pool, np.array([0, 1, 2 ** 32 - 1, 2 ** 32 - 2], np.uint32)]))
        rids, rindex = draw(rs, pool, nref, 0, hi, 0.1, 0.15)
        nq = int(rs.randint(1, 30))
        qids, qindex = draw(rs, pool, nq, 0, hi, 0.1, 0.15)
        if qids.size:
            qids[::13] ^= np.uint32(rs.randint(0, 4))              # ids outside the database
        ctx = np.diff(qindex).astype(np.uint32)
        if rs.rand() < 0.3:
            ctx[rs.randint(0, nq)] = 0                              # skipped whatever the list holds
        og, ori, ore = ob.mco_build(rids, rindex)
        want = ob.mco_count(og, ori, ore, qids, qindex, ctx, nref)
        g, ri, re_ = m.build(rids, rindex)
        assert np.array_equal(g, og) and np.array_equal(ri, ori) and np.array_equal(re_, ore), "build differs (it %d)" % it
        row0 = int(rs.randint(0, 2 ** 32 - 70000)) if ori.size == 0 or rs.rand() < 0.5 else max(0, int(ori[rs.randint(0, ori.size)]) - 300)
        nrows = int(min(rs.randint(1, 70000), 2 ** 32 - row0))
        ends = np.concatenate([[0], ore]).astype(np.uint64)
        wantrows = ends[np.searchsorted(ori.astype(np.uint64), np.arange(row0, row0 + nrows, dtype=np.uint64), side="right")]
        assert np.array_equal(m.index_rows(row0, nrows), wantrows), "index rows differ (it %d)" % it
        got = m.count(nref, qindex, ctx, [{"qry_ids": qids}])
        assert np.array_equal(got, want), "count (device row table) differs (it %d)" % it
        if ori.size:
            u = np.searchsorted(ori, qids, side="left")
            uc = np.minimum(u, ori.size - 1)
            hit = (u < ori.size) & (ori[uc] == qids)
            es = np.where(hit, ends[uc], 0).astype(np.uint64)
            ee = np.where(hit, ends[uc + 1], 0).astype(np.uint64)
            got = m.count(nref, qindex, ctx, [{"gids": og, "ext_start": es, "ext_end": ee}])
            assert np.array_equal(got, want), "count (host extents) differs (it %d)" % it
        incr += int(want.sum(dtype=np.uint64))
    print("fuzz_search: %d databases, %d increments checked, 0 mismatches (%.0f s, seed %d)" % (it, incr, time.time() - t0, args.seed))
    m.close()


if __name__ == "__main__":
    main()
