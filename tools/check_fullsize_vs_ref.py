#!/usr/bin/env python3
"""tools/check_fullsize_vs_ref.py -- BASELINE config 3 (and, with --reads 500000000, config 4's table regime) against the COMPILED
REFERENCE at full size.

The workload's reads (bench.py: seed 20261002, 150 bp, uniform) are written as one FASTQ file to /dev/shm; the compiled reference
(oracle/_ref/metakssd, built by oracle/Makefile straight from the reference's sources) sketches it with `dist -L L3K11.shuf -A -p
<cores>` (mt_shortreads2koc + write_fqkoc2files, iseq2comem.c:657-727, :516-562), the product command line sketches the same file
on the GPU, and the two sketch directories are compared as sorted (id, count) multisets per component: at -p > 1 the reference's
insertion order -- and with it the byte order of its files -- is not reproducible, the keys and their counts are.
The reference's check-then-write insert (iseq2comem.c:701-718) is racy under OpenMP: keys it LOSES or DOUBLES at this size are
reported, not hidden (`reference_only` / `product_only` / `count_differs`); `reference_p1_sample` adds an exact byte comparison
on a sample the reference finishes at -p 1.

    python tools/check_fullsize_vs_ref.py [--reads 50000000] [--p1-sample 2000000] [--out profiles/r04_fullsize_vs_reference.json]
Needs a GPU, oracle/_ref/metakssd and about 2.1 x reads x 320 bytes of /dev/shm."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED, READ_LEN = 20261002, 150


def load_dir(d):
    """-> list over components of sorted (id << 16 | count) arrays, and the component sizes in file order"""
    comps = []
    c = 0
    while os.path.exists(os.path.join(d, "combco.%d" % c)):
        ids = np.fromfile(os.path.join(d, "combco.%d" % c), dtype=np.uint32)
        cnt = np.fromfile(os.path.join(d, "combco.%d.a" % c), dtype=np.uint16)
        assert ids.size == cnt.size
        comps.append((ids, cnt))
        c += 1
    return comps


def compare(ref, prod):
    out = {"components": len(ref), "reference_keys": int(sum(i.size for i, _ in ref)), "product_keys": int(sum(i.size for i, _ in prod))}
    ref_only = prod_only = differs = 0
    same_bytes = len(ref) == len(prod)
    for (ri, rc), (pi, pc) in zip(ref, prod):
        same_bytes = same_bytes and np.array_equal(ri, pi) and np.array_equal(rc, pc)
        ro, po = np.argsort(ri, kind="stable"), np.argsort(pi, kind="stable")
        rs, ps = ri[ro], pi[po]
        ref_only += int(np.setdiff1d(rs, ps, assume_unique=False).size)
        prod_only += int(np.setdiff1d(ps, rs, assume_unique=False).size)
        common, ia, ib = np.intersect1d(rs, ps, assume_unique=False, return_indices=True)
        differs += int(np.count_nonzero(rc[ro][ia] != pc[po][ib]))
        out.setdefault("reference_duplicate_ids", 0)
        out["reference_duplicate_ids"] += int(rs.size - np.unique(rs).size)
    out.update({"reference_only": ref_only, "product_only": prod_only, "count_differs": differs,
                "multisets_equal": ref_only == 0 and prod_only == 0 and differs == 0 and out["reference_duplicate_ids"] == 0,
                "bytes_equal": bool(same_bytes)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--p1-sample", type=int, default=2_000_000, help="reads of a second file the reference also sketches at -p 1 (byte comparison); 0 = skip")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from metakssd_amd import capi
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "metakssd")
    cli = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
    if not os.path.exists(ref):
        sys.exit("oracle/_ref/metakssd is missing: `make -C oracle ref` where the reference's sources are")
    tmp = tempfile.mkdtemp(prefix="mkfull_", dir="/dev/shm")
    res = {"tool": "tools/check_fullsize_vs_ref.py", "reads": a.reads, "read_len": READ_LEN, "seed": SEED, "shuf": "L3K11 {k=11, subk=6, drlevel=3}, seed 11",
           "cores": cores}
    try:
        sp = os.path.join(tmp, "L3K11.shuf")
        capi.Shuf.generate(11, 6, 3, 11).write(sp)

        def run_pair(tag, reads, p):
            fq = os.path.join(tmp, tag + ".fq")
            assert capi.lib.mk_synth_fastq_write_mt(fq.encode(), SEED, 0, reads, READ_LEN, min(cores, 64)) == 0
            t0 = time.perf_counter()
            r = subprocess.run([ref, "dist", "-L", sp, "-A", "-p", str(p), "-o", os.path.join(tmp, tag + "_ref"), fq], stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE)
            t_ref = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError("reference failed: " + r.stderr.decode(errors="replace")[-400:])
            t0 = time.perf_counter()
            r = subprocess.run([cli, "dist", "-L", sp, "-A", "-o", os.path.join(tmp, tag + "_gpu"), "--quiet", fq], stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE)
            t_gpu = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError("product failed: " + r.stderr.decode(errors="replace")[-400:])
            os.unlink(fq)
            c = compare(load_dir(os.path.join(tmp, tag + "_ref")), load_dir(os.path.join(tmp, tag + "_gpu")))
            c.update({"reads": reads, "reference_threads": p, "reference_wall_s": round(t_ref, 2), "product_wall_s": round(t_gpu, 3),
                      "reference_gbases_s": round(reads * READ_LEN / t_ref / 1e9, 4), "file_gb": round(reads * (2 * READ_LEN + 18) / 1e9, 2)})
            return c
        res["full_size"] = run_pair("full", a.reads, cores)
        print(json.dumps(res["full_size"]), flush=True)
        if a.p1_sample:
            res["reference_p1_sample"] = run_pair("p1", a.p1_sample, 1)
            print(json.dumps(res["reference_p1_sample"]), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res["ok"] = bool(res["full_size"]["multisets_equal"] and (not a.p1_sample or res["reference_p1_sample"]["bytes_equal"]))
    print(json.dumps({"ok": res["ok"], "keys": res["full_size"]["product_keys"]}))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
