#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerates the golden vectors with the REAL reference.

Runs only in the build container (needs oracle/_ref/metakssd, compiled from /root/reference by
`make -C oracle ref`).  For every case it runs `metakssd dist -L <shuf> [-A] [-u] -p 1 -o out <input>` and stores
the payload files (combco.N, combco.N.a, combco.index.N) plus the decoded cofiles.stat fields under
tests/golden/expected/<case>/.  Inputs are either committed (tests/golden/inputs/*.gz, written here from seeded
numpy streams) or regenerated from a formula recorded in the manifest.  .shuf tables come from the product's
seeded generator; their sha256 is recorded so that a generator change cannot go unnoticed.

Nothing here is copied reference source: the outputs are data produced by executing the reference.
"""
import gzip
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import util_inputs as ui  # noqa: E402
from golden_cases import (CASES, COMPOSITE_CASES, CSZ6, SEARCH_CASES, SEARCH_DBS, SET_CASES, SHUF_SPECS, build_composite_inputs,  # noqa: E402
                          build_input, build_search_inputs, build_set_inputs, make_shuf)

REF = os.path.join(ROOT, "oracle", "_ref", "metakssd")
ORA = os.path.join(ROOT, "oracle", "kssd_oracle_cli")  # only to lay out sketch directories in a given order (set -g cases)


def parse_stat(path):
    b = open(path, "rb").read()
    shuf_id, koc = struct.unpack_from("<IB", b, 0)
    kmerlen, dim_rd_len, comp_num, infile_num, all_ctx = struct.unpack_from("<iiiiQ", b, 8)
    cts = list(struct.unpack_from("<%dI" % infile_num, b, 32))
    return dict(shuf_id=shuf_id, koc=koc, kmerlen=kmerlen, dim_rd_len=dim_rd_len, comp_num=comp_num,
                infile_num=infile_num, all_ctx_ct=all_ctx, ctx_ct=cts)


def tree_section():
    """every committed fixture file (expected/ and inputs/) -> sha256: the tests take the expected file set from here, never
    from os.listdir, so a fixture that goes missing (e.g. swallowed by a .gitignore pattern) turns the suite red"""
    tree = {}
    for sub in ("expected", "inputs"):
        for dp, _, fs in os.walk(os.path.join(HERE, sub)):
            for f in fs:
                p = os.path.join(dp, f)
                tree[os.path.relpath(p, HERE)] = hashlib.sha256(open(p, "rb").read()).hexdigest()
    return tree


def search_section(manifest, work, shuf_paths, exp_root):
    """stage II + `dist -r`: every database costs one 32 GiB mco.index.0 (deleted again); its xxh64 is what is kept"""
    import xxhash
    manifest["search_dbs"], manifest["search_cases"] = {}, {}
    for db, c in SEARCH_DBS.items():
        refs = build_search_inputs(db, c["refs"], work, write_committed=True)
        sk, mco = db + ".sk", db + ".mco"
        for cmd in ([ORA, "-L", shuf_paths[c["shuf"]], "-o", sk] + refs, [REF, "dist", "-o", mco, sk]):
            r = subprocess.run(cmd, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if r.returncode != 0:
                raise SystemExit("%s: step failed: %s\n%s" % (db, " ".join(cmd), r.stderr.decode(errors="replace")[-300:]))
        d = os.path.join(exp_root, db)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        for f in ("mcofiles.stat", "mco.0"):
            shutil.copy(os.path.join(work, mco, f), os.path.join(d, f))
        h = xxhash.xxh64()
        with open(os.path.join(work, mco, "mco.index.0"), "rb") as f:
            while True:
                b = f.read(1 << 26)
                if not b:
                    break
                h.update(b)
        manifest["search_dbs"][db] = {"shuf": c["shuf"], "refs": c["refs"], "index_xxh64": h.hexdigest(),
                                      "index_bytes": os.path.getsize(os.path.join(work, mco, "mco.index.0")),
                                      "gids": os.path.getsize(os.path.join(d, "mco.0")) // 4}
        print("%-28s gids=%d index xxh64=%s" % (db, manifest["search_dbs"][db]["gids"], h.hexdigest()))
        for case, sc in SEARCH_CASES.items():
            if sc["db"] != db:
                continue
            qry = build_search_inputs(case, sc["query"], work, write_committed=True)
            qsk, out = case + ".qsk", case + ".out"
            for cmd in ([ORA, "-L", shuf_paths[c["shuf"]]] + sc["qflags"] + ["-o", qsk] + qry,
                        [REF, "dist", "-r", mco, "-o", out] + sc["flags"] + ["--keepskf", qsk]):
                r = subprocess.run(cmd, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                if r.returncode != 0:
                    raise SystemExit("%s: step failed: %s\n%s" % (case, " ".join(cmd), r.stderr.decode(errors="replace")[-300:]))
            dc = os.path.join(exp_root, case)
            shutil.rmtree(dc, ignore_errors=True)
            os.makedirs(dc)
            for f in ("distance.out", "sharedk_ct.dat"):
                shutil.copy(os.path.join(work, out, f), os.path.join(dc, f))
            nlines = len(open(os.path.join(dc, "distance.out")).read().splitlines())
            manifest["search_cases"][case] = dict(sc, lines=nlines)
            print("%-28s lines=%d" % (case, nlines))
        os.remove(os.path.join(work, mco, "mco.index.0"))


def csz6_section(manifest, work, exp_root):
    """a 16-component database end to end by the reference built with -DCOMPONENT_SZ=6: stage I per genome (-p 1), the sketch
    directories strung together by its combine_queries (a fixed order: a multi-file stage I shuffles by the clock), stage II,
    two searches"""
    ref6 = os.path.join(ROOT, "oracle", "_ref", "metakssd_csz6")
    sp = os.path.join(work, CSZ6["shuf"] + ".shuf")
    make_shuf(CSZ6["shuf"], sp)
    manifest["shufs"][CSZ6["shuf"]] = {"spec": SHUF_SPECS[CSZ6["shuf"]], "sha256": hashlib.sha256(open(sp, "rb").read()).hexdigest()}

    def run(cmd):
        r = subprocess.run(cmd, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise SystemExit("csz6: step failed: %s\n%s" % (" ".join(cmd), r.stderr.decode(errors="replace")[-300:]))

    def sketch_all(tag, specs):
        files = build_search_inputs("csz6_" + tag, specs, work, write_committed=True)
        dirs = []
        for i, f in enumerate(files):
            d = "csz6_%s_%d.sk" % (tag, i)
            run([ref6, "dist", "-L", sp, "-p", "1", "-o", d, f])
            dirs.append(d)
        run([ref6, "dist", "-o", "csz6_%s.sk" % tag] + dirs)
        return "csz6_%s.sk" % tag, files

    refsk, ref_files = sketch_all("ref", CSZ6["refs"])
    qsk, q_files = sketch_all("qry", CSZ6["query"])
    # stage II: the reference's combco2mco() frees every row pointer of the component table after each component, also rows
    # it did not allocate for THAT component (co2mco.c:22-23, 74-80: stale pointers of the component before) -- glibc aborts
    # with "double free detected" on the second component.  Recorded, not worked around: there is no reference output for a
    # multi-component mco database; the product's is checked per component against the oracle's restatement instead.
    r = subprocess.run([ref6, "dist", "-o", "csz6.mco", refsk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    stage2_crash = r.returncode != 0
    d = os.path.join(exp_root, "csz6_db")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, "ref_sk")); os.makedirs(os.path.join(d, "qry_sk"))
    entry = {"shuf": CSZ6["shuf"], "component_sz": 6, "ref_files": ref_files, "qry_files": q_files,
             "reference_stage2_aborts": stage2_crash, "reference_stage2_stderr": r.stderr.decode(errors="replace")[-120:].strip()}
    for sub, src in (("ref_sk", refsk), ("qry_sk", qsk)):
        for f in sorted(os.listdir(os.path.join(work, src))):
            if f.startswith("combco"):
                shutil.copy(os.path.join(work, src, f), os.path.join(d, sub, f))
        entry[sub + "_stat"] = parse_stat(os.path.join(work, src, "cofiles.stat"))
    assert entry["ref_sk_stat"]["comp_num"] == 16
    manifest["csz6"] = entry
    print("csz6_db: 16 components, %d reference ids, %d query ids; reference stage II aborts: %s (%s)" % (
        entry["ref_sk_stat"]["all_ctx_ct"], entry["qry_sk_stat"]["all_ctx_ct"], stage2_crash, entry["reference_stage2_stderr"]))


def main():
    if "--tree-only" in sys.argv:  # re-pin the committed fixture tree; refuses files whose per-case hash (made with the
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))  # reference at hand) no longer matches
        tree = tree_section()
        for sec in ("cases", "set_cases"):
            for case, e in manifest.get(sec, {}).items():
                if isinstance(e.get("files"), dict):
                    for f, h in e["files"].items():
                        if tree.get(os.path.join("expected", case, f)) != h:
                            raise SystemExit("%s/%s: committed file is missing or differs from the reference-made hash" % (case, f))
        manifest["tree"] = tree
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        print("tree: %d files pinned" % len(tree))
        return
    if "--only-csz6" in sys.argv:
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        work = tempfile.mkdtemp(prefix="golden_")
        csz6_section(manifest, work, os.path.join(HERE, "expected"))
        manifest["tree"] = tree_section()
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        shutil.rmtree(work, ignore_errors=True)
        return
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/metakssd missing: run `make -C oracle ref`")
    exp_root = os.path.join(HERE, "expected")
    if "--only-search" in sys.argv:   # add / refresh the stage II + search vectors, leave the rest as it is
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        work = tempfile.mkdtemp(prefix="golden_")
        shuf_paths = {}
        for name in sorted({c["shuf"] for c in SEARCH_DBS.values()}):
            shuf_paths[name] = os.path.join(work, name + ".shuf")
            make_shuf(name, shuf_paths[name])
            assert manifest["shufs"][name]["sha256"] == hashlib.sha256(open(shuf_paths[name], "rb").read()).hexdigest()
        search_section(manifest, work, shuf_paths, exp_root)
        manifest["tree"] = tree_section()
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        shutil.rmtree(work, ignore_errors=True)
        return
    shutil.rmtree(exp_root, ignore_errors=True)
    os.makedirs(exp_root)
    os.makedirs(os.path.join(HERE, "inputs"), exist_ok=True)
    work = tempfile.mkdtemp(prefix="golden_")
    manifest = {"shufs": {}, "cases": {}}
    shuf_paths = {}
    for name in sorted({c["shuf"] for c in CASES.values()} | {c["shuf"] for c in SET_CASES.values()} |
                       {c["shuf"] for c in COMPOSITE_CASES.values()} | {c["shuf"] for c in SEARCH_DBS.values()}):
        p = os.path.join(work, name + ".shuf")
        make_shuf(name, p)
        shuf_paths[name] = p
        manifest["shufs"][name] = {"spec": SHUF_SPECS.get(name), "sha256": hashlib.sha256(open(p, "rb").read()).hexdigest()}
    for case, c in CASES.items():
        inp = build_input(case, work, write_committed=True)
        out = os.path.join(work, case + ".out")
        cmd = [REF, "dist", "-L", shuf_paths[c["shuf"]]] + c["flags"] + ["-p", "1", "-o", out, inp]
        r = subprocess.run(cmd, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        err = r.stderr.decode(errors="replace")
        aborted = "too crowd" in err and not os.path.exists(os.path.join(out, "cofiles.stat"))
        entry = {"shuf": c["shuf"], "flags": c["flags"], "input": c["input"], "aborted": aborted}
        if not aborted:
            if r.returncode != 0:
                raise SystemExit("reference failed on %s: %s" % (case, err[-300:]))
            d = os.path.join(exp_root, case)
            os.makedirs(d)
            for f in sorted(os.listdir(out)):
                if f.startswith("combco"):
                    shutil.copy(os.path.join(out, f), os.path.join(d, f))
            entry["stat"] = parse_stat(os.path.join(out, "cofiles.stat"))
            entry["files"] = {f: hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest() for f in sorted(os.listdir(d))}
        manifest["cases"][case] = entry
        print("%-28s %s" % (case, "ABORT (too crowd)" if aborted else "distinct=%d" % entry["stat"]["all_ctx_ct"]))
    # ---- `set -u` / `set -q`: reference dist makes the sketch directory, reference set makes pan.N / uniq_pan.N ----
    manifest["set_cases"] = {}
    for case, c in SET_CASES.items():
        inputs = build_set_inputs(case, work, write_committed=True)
        sk, out = os.path.join(work, case + ".sk"), os.path.join(work, case + ".pan")
        if c["op"] == "-g":
            r = subprocess.run([ORA, "-L", shuf_paths[c["shuf"]]] + c["flags"] + ["-o", sk] + inputs, cwd=work,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            taxf = os.path.join(work, case + ".tsv")
            open(taxf, "w").write("".join(t + "\n" for t in c["tax"]))
            r = subprocess.run([REF, "set", "-g", taxf, "-o", out, sk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if r.returncode != 0 or not os.path.exists(os.path.join(out, "cofiles.stat")):
                raise SystemExit("reference set -g failed on %s: %s" % (case, r.stderr.decode(errors="replace")[-300:]))
            d = os.path.join(exp_root, case)
            os.makedirs(d)
            for f in sorted(os.listdir(out)):
                if f.startswith("combco"):
                    shutil.copy(os.path.join(out, f), os.path.join(d, f))
            b = open(os.path.join(out, "cofiles.stat"), "rb").read()
            st = parse_stat(os.path.join(out, "cofiles.stat"))
            n = st["infile_num"]
            names = [b[32 + 4 * n + 256 * i: 32 + 4 * n + 256 * (i + 1)].split(b"\0", 1)[0].decode() for i in range(n)]
            manifest["set_cases"][case] = {
                "shuf": c["shuf"], "flags": c["flags"], "inputs": c["inputs"], "op": "-g", "tax": c["tax"], "stat": st, "names": names,
                "files": {f: hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest() for f in sorted(os.listdir(d))},
                "ids": sum(os.path.getsize(os.path.join(d, f)) // 4 for f in os.listdir(d) if ".index." not in f)}
            print("%-28s ids=%d taxa=%s" % (case, manifest["set_cases"][case]["ids"], names))
            continue
        # The reference's dist permutes its input list with a time-seeded shuffle (command_dist.c:215,
        # command_shuffle.c:143): fine for -u / -q (order-free), but -i / -s write one block per sketch in directory
        # order, so THEIR sketch directory is laid out by the pinned oracle CLI in the given order (as for -g) and only
        # the set operation itself is the reference's.
        dist_bin = [ORA] if "pan" in c else [REF, "dist", "-p", "1"]
        r = subprocess.run(dist_bin + ["-L", shuf_paths[c["shuf"]]] + c["flags"] + ["-o", sk] + inputs,
                           cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if not os.path.exists(os.path.join(sk, "cofiles.stat")):
            raise SystemExit("reference dist failed on %s: %s" % (case, r.stderr.decode(errors="replace")[-300:]))
        op_args = [c["op"]]
        if "pan" in c:  # the pan directory: reference dist on the pan inputs, then reference set -u/-q
            pc = c["pan"]
            pin = build_set_inputs(case, work, write_committed=True, pan=True)
            psk, pdir = os.path.join(work, case + ".psk"), os.path.join(work, case + ".pdir")
            subprocess.run([REF, "dist", "-L", shuf_paths[c["shuf"]]] + pc["flags"] + ["-p", "1", "-o", psk] + pin, cwd=work,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            subprocess.run([REF, "set", pc["op"], "-o", pdir, psk], cwd=work, input=b"N\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if not os.path.exists(os.path.join(pdir, "cofiles.stat")):
                raise SystemExit("reference pan step failed on %s" % case)
            op_args = [c["op"], pdir]
        r = subprocess.run([REF, "set"] + op_args + ["-o", out, sk], cwd=work, input=b"N\n", stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE)
        if r.returncode != 0 or not os.path.exists(os.path.join(out, "cofiles.stat")):
            raise SystemExit("reference set failed on %s: %s" % (case, r.stderr.decode(errors="replace")[-300:]))
        d = os.path.join(exp_root, case)
        os.makedirs(d)
        for f in sorted(os.listdir(out)):
            if f.startswith("pan.") or f.startswith("uniq_pan.") or f.startswith("combco"):
                shutil.copy(os.path.join(out, f), os.path.join(d, f))
        stat = parse_stat(os.path.join(sk, "cofiles.stat"))
        entry = {"shuf": c["shuf"], "flags": c["flags"], "inputs": c["inputs"], "op": c["op"],
                 "header": {k: stat[k] for k in ("shuf_id", "koc", "kmerlen", "dim_rd_len", "comp_num", "infile_num", "all_ctx_ct")},
                 "files": {f: hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest() for f in sorted(os.listdir(d))},
                 "ids": sum(os.path.getsize(os.path.join(d, f)) // 4 for f in os.listdir(d) if ".index." not in f)}
        if "pan" in c:
            entry["pan"] = c["pan"]
            entry["stat"] = parse_stat(os.path.join(out, "cofiles.stat"))  # full stat file: recounted ctx_ct, old all_ctx_ct
        manifest["set_cases"][case] = entry
        print("%-28s ids=%d files=%d" % (case, entry["ids"], len(entry["files"])))
    # ---- composite: marker database by the reference's set -g / -q / -i, then the reference's composite -q (text and -b) ----
    manifest["composite_cases"] = {}
    for case, c in COMPOSITE_CASES.items():
        refs, qry = build_composite_inputs(case, work, write_committed=True)
        sk, grp, uq, db, qsk = (os.path.join(work, case + sfx) for sfx in (".sk", ".grp", ".uq", ".db", ".qsk"))
        taxf = os.path.join(work, case + ".tsv")
        open(taxf, "w").write("".join(t + "\n" for t in c["tax"]))
        steps = [[ORA, "-L", shuf_paths[c["shuf"]], "-o", sk] + refs,
                 [REF, "set", "-g", taxf, "-o", grp, sk], [REF, "set", "-q", "-o", uq, grp], [REF, "set", "-i", uq, "-o", db, grp],
                 [ORA, "-L", shuf_paths[c["shuf"]], "-A", "-o", qsk] + qry]
        for cmd in steps:
            r = subprocess.run(cmd, cwd=work, input=b"N\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if r.returncode != 0:
                raise SystemExit("%s: step failed: %s\n%s" % (case, " ".join(cmd), r.stderr.decode(errors="replace")[-300:]))
        r = subprocess.run([REF, "composite", "-r", db, "-q", qsk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise SystemExit("reference composite failed on %s" % case)
        lines = []
        for ln in r.stdout.decode().splitlines():
            f = ln.split("\t")
            f[0] = os.path.basename(f[0])  # the query's path is whatever the test's tmp dir is
            lines.append("\t".join(f))
        d = os.path.join(exp_root, case)
        os.makedirs(d)
        open(os.path.join(d, "composite.tsv"), "w").write("".join(x + "\n" for x in lines))
        abv = os.path.join(work, case + ".abv")
        r = subprocess.run([REF, "composite", "-r", db, "-q", qsk, "-b", "-o", abv], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        for f in sorted(os.listdir(abv)):
            shutil.copy(os.path.join(abv, f), os.path.join(d, f))
        manifest["composite_cases"][case] = {"shuf": c["shuf"], "refs": c["refs"], "tax": c["tax"], "query": c["query"], "lines": len(lines),
                                             "files": sorted(os.listdir(d))}
        print("%-28s lines=%d files=%s" % (case, len(lines), sorted(os.listdir(d))))
        for x in lines[:3]:
            print("    " + x)
    search_section(manifest, work, shuf_paths, exp_root)
    manifest["tree"] = tree_section()
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(work, ignore_errors=True)
    tot = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(HERE) for f in fs)
    print("golden dir: %.1f KiB" % (tot / 1024))


if __name__ == "__main__":
    main()
