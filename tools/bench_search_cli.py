#!/usr/bin/env python3
"""tools/bench_search_cli.py -- the command line end to end on a synthetic database: writes a sketch directory of R genomes x
about G ids (and a query directory of Q sketches taken from it) in the reference's format, then times
`metakssd dist -o <mco> <sketches>` (stage II: mco.N + the 32 GiB mco.index.N on the scratch disk) and
`metakssd dist -r <mco> -o <out> <queries>` (index gather on the host, counting on the device, distance.out).

    python tools/bench_search_cli.py [--refs 2000] [--ids 20000] [--queries 200] [--dir /tmp/mk_search_bench]"""
import argparse
import json
import os
import shutil
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")


def write_sketch_dir(path, parts, prefix):
    os.makedirs(path)
    n = len(parts)
    sizes = np.array([p.size for p in parts], np.uint32)
    with open(os.path.join(path, "cofiles.stat"), "wb") as f:   # co_dstat_t, global_basic.h:116-126 (L3K10: k = 10, level 3)
        f.write(struct.pack("<IB3xiiiiQ", 4242, 0, 20, 6, 1, n, int(sizes.sum())))
        f.write(sizes.tobytes())
        for i in range(n):
            f.write(("%s_%06d.fa" % (prefix, i)).encode().ljust(256, b"\0"))
    np.concatenate(parts).astype(np.uint32).tofile(os.path.join(path, "combco.0"))
    np.concatenate([[0], np.cumsum(sizes, dtype=np.uint64)]).astype(np.uint64).tofile(os.path.join(path, "combco.index.0"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs", type=int, default=2000)
    ap.add_argument("--ids", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=200)
    ap.add_argument("--dir", default="/tmp/mk_search_bench")
    ap.add_argument("--ref-timeout", type=int, default=150, help="seconds allowed per reference step")
    ap.add_argument("--reference", action="store_true", help="also time the compiled reference (oracle/_ref/metakssd) on the same "
                    "directories and compare its files with ours; needs > 100 GB of free memory and 64 GiB more disk")
    a = ap.parse_args()
    if shutil.disk_usage(os.path.dirname(a.dir) or "/").free < 48 << 30:
        raise SystemExit("needs 32 GiB of scratch space for mco.index.0")
    shutil.rmtree(a.dir, ignore_errors=True)
    os.makedirs(a.dir)
    rs = np.random.RandomState(3)
    universe = np.unique(rs.randint(0, 2 ** 28, size=a.refs * 40 + 4 * a.ids, dtype=np.uint64).astype(np.uint32))   # L3K10 ids: 28 bits
    sizes = rs.randint(a.ids // 2, a.ids + a.ids // 2, size=a.refs)
    starts = np.sort(rs.randint(0, universe.size - 2 * a.ids, size=a.refs))
    parts = [rs.permutation(universe[s:s + n]) for s, n in zip(starts, sizes)]
    sel = rs.randint(0, a.refs, size=a.queries)
    write_sketch_dir(os.path.join(a.dir, "ref.sk"), parts, "genome")
    write_sketch_dir(os.path.join(a.dir, "qry.sk"), [parts[i] for i in sel], "query")
    out = {"refs": a.refs, "ref_ids": int(sizes.sum()), "queries": a.queries, "query_ids": int(sizes[sel].sum())}
    t0 = time.perf_counter()
    subprocess.run([CLI, "dist", "--quiet", "-o", "ref.mco", "ref.sk"], cwd=a.dir, check=True, stdin=subprocess.DEVNULL)
    out["stage2_s"] = time.perf_counter() - t0
    out["index_bytes"] = os.path.getsize(os.path.join(a.dir, "ref.mco", "mco.index.0"))
    t0 = time.perf_counter()
    subprocess.run([CLI, "dist", "--quiet", "-p", "16", "-r", "ref.mco", "-o", "hits", "-N", "5", "--keepskf", "qry.sk"], cwd=a.dir, check=True, stdin=subprocess.DEVNULL)
    out["search_s"] = time.perf_counter() - t0
    ct = np.fromfile(os.path.join(a.dir, "hits", "sharedk_ct.dat"), np.uint32).reshape(a.queries, a.refs)
    assert all(ct[k, sel[k]] == sizes[sel[k]] for k in range(a.queries))            # every query finds itself completely
    out["increments"] = int(ct.sum(dtype=np.uint64))
    out["distance_lines"] = sum(1 for _ in open(os.path.join(a.dir, "hits", "distance.out")))
    ref = os.path.join(ROOT, "oracle", "_ref", "metakssd")
    if a.reference and os.path.exists(ref):
        import filecmp
        free_gb = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) / 1e6
        if free_gb < 100 or shutil.disk_usage(a.dir).free < 40 << 30:
            out["reference"] = "skipped: %.0f GB of memory available, %d GiB of disk" % (free_gb, shutil.disk_usage(a.dir).free >> 30)
        else:
            cores = os.cpu_count()
            res = {"cores": cores}
            try:   # every reference step is bounded: at 400 M ids its stage II alone ran past 20 minutes
                t0 = time.perf_counter()
                subprocess.run([ref, "dist", "-p", str(cores), "-o", "ref.mco2", "ref.sk"], cwd=a.dir, check=True, stdin=subprocess.DEVNULL,
                               stdout=subprocess.DEVNULL, timeout=a.ref_timeout)
                res["stage2_s"] = time.perf_counter() - t0
                t0 = time.perf_counter()
                subprocess.run([ref, "dist", "-p", str(cores), "-r", "ref.mco2", "-o", "hits2", "-N", "5", "--keepskf", "qry.sk"], cwd=a.dir,
                               check=True, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, timeout=a.ref_timeout)
                res["search_s"] = time.perf_counter() - t0
                same = {f: filecmp.cmp(os.path.join(a.dir, "ref.mco", f), os.path.join(a.dir, "ref.mco2", f), shallow=False)
                        for f in ("mco.0", "mcofiles.stat", "mco.index.0")}
                same.update({f: filecmp.cmp(os.path.join(a.dir, "hits", f), os.path.join(a.dir, "hits2", f), shallow=False)
                             for f in ("sharedk_ct.dat", "distance.out")})
                res["files_identical"] = same
            except subprocess.TimeoutExpired as e:
                res["timed_out"] = "%s after %d s" % (" ".join(e.cmd[1:4]), a.ref_timeout)
            out["reference"] = res
    shutil.rmtree(a.dir, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
