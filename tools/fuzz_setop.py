#!/usr/bin/env python3
"""tools/fuzz_setop.py -- randomized differential test of mk_setop_* (set -u/-q/-i/-s/-g, composite join) against numpy's
definition of the same sets and the oracle's per-taxon table (GPU box only).   python tools/fuzz_setop.py [--cases 300] [--seed 1]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from metakssd_amd import capi
    from oracle_binding import load
    lib, ora = capi.lib, load()
    ora.ko_group_layout.restype = C.c_size_t
    ora.ko_group_layout.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
    lib.mk_setop_filter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_uint64), C.c_void_p]
    lib.mk_setop_group.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.mk_setop_join.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    lib.mk_setop_group_table_size.restype = C.c_uint32
    lib.mk_setop_group_table_size.argtypes = [C.c_uint64]
    so = capi.SetOp(0)
    sizes = [0, 1, 2, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 2047, 4096, 16383, 16385, 70000, 300000]

    def ids_of(rs, n):
        kind = rs.randint(0, 4)
        if kind == 0:
            v = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)
        elif kind == 1:
            v = rs.randint(0, max(2, n // 3 + 1), size=n, dtype=np.uint64)             # many duplicates, small values, zeros
        elif kind == 2:
            v = rs.randint(2 ** 32 - 5000, 2 ** 32, size=n, dtype=np.uint64)           # top of the range
        else:
            v = (rs.randint(0, 4000, size=n, dtype=np.uint64) * 1048583) % 2 ** 32      # spread, moderately repeated
        return v.astype(np.uint32)

    def out_array(ptr, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    bad = 0
    for case in range(a.cases):
        rs = np.random.RandomState(a.seed * 7919 + case)
        op = ["union", "uniq", "filter", "group", "join"][rs.randint(0, 5)]
        n = int(rs.choice(sizes))
        ids = ids_of(rs, n)
        ok, desc = True, "case %d seed %d %s n=%d" % (case, a.seed, op, n)
        if op in ("union", "uniq"):
            parts = np.array_split(ids, int(rs.choice([1, 2, 5]))) if n else [ids]
            got = so.union(parts, uniq=op == "uniq")
            vals, cnt = np.unique(ids, return_counts=True)
            ok = np.array_equal(got, vals[cnt == 1] if op == "uniq" else vals)
        elif op == "filter":
            pan = ids_of(rs, int(rs.choice(sizes[:14])))
            keep = int(rs.randint(0, 2))
            cuts = np.unique(np.concatenate([[0, n], rs.randint(0, n + 1, size=int(rs.choice([0, 1, 9])))])).astype(np.uint64)
            bout = np.zeros(cuts.size, np.uint64)
            so.begin(uniq=False)
            if pan.size:
                assert lib.mk_setop_add(so.h, pan.ctypes.data, pan.size) == 0
            o, m = C.c_void_p(), C.c_uint64(0)
            assert lib.mk_setop_filter(so.h, keep, ids.ctypes.data if n else None, n, cuts.ctypes.data, cuts.size, C.byref(o), C.byref(m),
                                       bout.ctypes.data) == 0
            member = np.isin(ids, pan)
            kept = member if keep else ~member
            ok = np.array_equal(out_array(o, m.value), ids[kept]) and \
                np.array_equal(bout, np.concatenate([[0], np.cumsum(kept)])[cuts.astype(np.int64)].astype(np.uint64))
        elif op == "group":
            S = lib.mk_setop_group_table_size(max(1, n)) if rs.rand() < 0.8 else int(rs.choice([251, 509, 1021, 4093]))
            o, m = C.c_void_p(), C.c_uint64(0)
            assert lib.mk_setop_group(so.h, ids.ctypes.data if n else None, n, S, C.byref(o), C.byref(m)) == 0
            want = np.zeros(max(1, n), np.uint32)
            k = ora.ko_group_layout(ids.ctypes.data if n else None, n, S, want.ctypes.data)
            ok = np.array_equal(out_array(o, m.value), want[:k])
            desc += " S=%d" % S
        else:
            q = np.unique(ids_of(rs, int(rs.choice(sizes[:16]))))
            rs.shuffle(q)
            qab = rs.randint(1, 65536, size=q.size).astype(np.uint16)
            cuts = np.unique(np.concatenate([[0, n], rs.randint(0, n + 1, size=int(rs.choice([0, 3, 50])))])).astype(np.uint64)
            bout = np.zeros(cuts.size, np.uint64)
            o, m = C.c_void_p(), C.c_uint64(0)
            assert lib.mk_setop_join(so.h, q.ctypes.data if q.size else None, qab.ctypes.data if q.size else None, q.size,
                                     ids.ctypes.data if n else None, n, cuts.ctypes.data, cuts.size, C.byref(o), C.byref(m), bout.ctypes.data) == 0
            order = np.argsort(q, kind="stable")
            qs, abs_ = q[order], qab[order]
            pos = np.searchsorted(qs, ids)
            pos[pos >= qs.size] = 0
            hit = (qs[pos] == ids) if qs.size else np.zeros(n, bool)
            ok = np.array_equal(out_array(o, m.value), abs_[pos[hit]].astype(np.uint32)) and \
                np.array_equal(bout, np.concatenate([[0], np.cumsum(hit)])[cuts.astype(np.int64)].astype(np.uint64))
            desc += " q=%d" % q.size
        if not ok:
            bad += 1
            print("MISMATCH", desc)
            break
    print("%d cases, %d mismatches" % (case + 1, bad))
    so.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
