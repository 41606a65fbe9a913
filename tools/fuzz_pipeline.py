#!/usr/bin/env python3
"""tools/fuzz_pipeline.py -- the README's MarkerDB recipe and the profiling step, end to end, on random inputs: the product command
line (GPU) against the pinned oracle command line + the COMPILED REFERENCE (oracle/_ref/metakssd), stage by stage.

    sketch directory of several genome files      product: metakssd dist -L .. -o sk refs..        checker: kssd_oracle_cli (given order)
    set -g tax / set -q / set -i                   product: metakssd set ..                         checker: oracle/_ref/metakssd set ..
    query sketch (-A) of one or more FASTQ files   product: metakssd dist -A ..                     checker: kssd_oracle_cli -A
    composite -r db -q qsk  (text and -b)          product: metakssd composite ..                   checker: oracle/_ref/metakssd composite ..

Every stage's output directory is compared file by file (cofiles.stat field-wise: the reference leaves padding bytes unset).
A case that differs is kept whole under --keep (default gpurun_out/fuzz_pipeline_fail/<case>/) with the commands, return codes and
stderr of every process.  MK_POISON (host/mk_host_internal.h) is passed to the PRODUCT's processes only (--poison).

    python tools/fuzz_pipeline.py --cases 3000 --workers 24 --poison 0xA5 [--l2k11 0.15] [--seed 1]
    python tools/fuzz_pipeline.py --golden composite_mix_L2K11 --times 300 --workers 4    # one golden pipeline again and again
"""
import argparse
import concurrent.futures as cf
import filecmp
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util_inputs as ui  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "metakssd")
ORA = os.path.join(ROOT, "oracle", "kssd_oracle_cli")
CLI = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
# (k, subk, drlevel): small tables with collisions, one component; L3K10 (config 5); L2K11 (config 5: 16 components, 537 M slots)
GEOM_SMALL = [(7, 4, 1), (6, 3, 0), (8, 4, 2), (9, 5, 2), (10, 6, 3)]
GEOM_L2K11 = (11, 5, 2)


def parse_stat(path):
    b = open(path, "rb").read()
    n = struct.unpack_from("<i", b, 20)[0]
    names = [b[32 + 4 * n + 256 * i:32 + 4 * n + 256 * (i + 1)].split(b"\0", 1)[0] for i in range(n)]
    return b[:5], b[8:32 + 4 * n], [os.path.basename(x) for x in names]


def same_dir(a, b):
    """None when the two directories hold the same files with the same bytes, else what differs"""
    la, lb = sorted(os.listdir(a)), sorted(os.listdir(b))
    if la != lb:
        return "listing %s != %s" % (la, lb)
    for f in la:
        pa, pb = os.path.join(a, f), os.path.join(b, f)
        if f in ("cofiles.stat",):
            if parse_stat(pa) != parse_stat(pb):
                return "cofiles.stat fields"
        elif not filecmp.cmp(pa, pb, shallow=False):
            return "%s (%d vs %d bytes)" % (f, os.path.getsize(pa), os.path.getsize(pb))
    return None


def run(cmd, cwd, log, env=None, stdin=b"N\n"):
    r = subprocess.run(cmd, cwd=cwd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    log.append({"cmd": [os.path.relpath(c, ROOT) if c.startswith(ROOT) else c for c in cmd], "rc": r.returncode,
                "stderr": r.stderr.decode(errors="replace")[-600:]})
    return r


def norm_composite(text):
    out = []
    for ln in text.splitlines():
        f = ln.split("\t")
        f[0] = os.path.basename(f[0])
        out.append("\t".join(f))
    return out


def make_case(rs, work, l2k11):
    """random strain files, taxonomy and query reads; returns (geometry, refs, tax lines, query files)"""
    k, subk, drl = GEOM_L2K11 if l2k11 else GEOM_SMALL[rs.randint(0, len(GEOM_SMALL))]
    dense = 16 ** (subk - drl) >= 16 ** subk  # accept-everything tables: keep the inputs small (131 071 slots)
    glen = int(rs.choice([3000, 8000] if dense else [20000, 60000, 150000]))
    g = ui.rand_seq(rs, glen)
    nref = int(rs.randint(2, 7))
    refs, strains = [], []
    for i in range(nref):
        a = int(rs.randint(0, glen // 2))
        b = int(rs.randint(a + glen // 4, glen + 1))
        s = bytearray(g[a:b])
        for _ in range(int(rs.randint(0, 1 + len(s) // 2000))):  # point mutations: private k-mers
            s[rs.randint(0, len(s))] = b"ACGT"[rs.randint(0, 4)]
        if rs.rand() < 0.2:
            p = int(rs.randint(0, len(s)))
            s[p:p + int(rs.randint(1, 40))] = b"N" * int(rs.randint(1, 40))
        s = bytes(s)
        if rs.rand() < 0.2:
            s = s.lower()
        strains.append(s.upper().replace(b"N", b"A"))
        ncontig = int(rs.randint(1, 4))
        cuts = sorted(int(x) for x in rs.randint(0, len(s) + 1, ncontig - 1))
        contigs = [s[x:y] for x, y in zip([0] + cuts, cuts + [len(s)])]
        if rs.rand() < 0.1:
            contigs.append(b"")
        width = int(rs.choice([60, 70, 80, 1000000]))
        text = b"".join(b">c%d some text\n" % j + b"".join(c[x:x + width] + b"\n" for x in range(0, len(c), width)) for j, c in enumerate(contigs))
        if rs.rand() < 0.15:
            text = text.replace(b"\n", b"\r\n")
        p = os.path.join(work, "ref%d.fa" % i)
        open(p, "wb").write(text)
        refs.append(p)
    taxa = [(11, "eleven"), (12, "twelve"), (13, ""), (0, "left out"), (28901, "Salmonella enterica")]
    tax = []
    for i in range(nref):
        t, name = taxa[rs.randint(0, len(taxa))]
        tax.append("%d\t%s" % (t, name) if name else "%d" % t)
    if all(t.startswith("0") for t in tax):
        tax[0] = "11\televen"
    qfiles = []
    for q in range(int(rs.randint(1, 4))):
        nreads = int(rs.choice([40, 300] if dense else [300, 3000, 12000]))
        seqs = []
        for _ in range(nreads):
            s = strains[rs.randint(0, nref)]
            L = int(rs.choice([60, 100, 150, 150, 150, 250]))
            if len(s) <= L:
                continue
            a = int(rs.randint(0, len(s) - L))
            r = s[a:a + L]
            seqs.append(ui.revcomp(r) if rs.rand() < 0.5 else r)
        p = os.path.join(work, "qry%d.fq" % q)
        open(p, "wb").write(ui.fastq_bytes(seqs))
        qfiles.append(p)
    return (k, subk, drl), refs, tax, qfiles


KEPT = [0]
KEEP_FULL = 6  # cases kept with their files (gpurun brings at most 64 MiB home); every further one keeps its log only


def keep_case(keep, name, work, info):
    import threading
    dst = os.path.join(keep, name)
    shutil.rmtree(dst, ignore_errors=True)
    with keep_case.lock:
        KEPT[0] += 1
        full = KEPT[0] <= KEEP_FULL
    if full:  # everything but files above 2 MiB (the inputs of the large geometries: the case is reproducible from its seed)
        for d, _, files in os.walk(work):
            for fn in files:
                src = os.path.join(d, fn)
                if os.path.getsize(src) <= (2 << 20):
                    out = os.path.join(dst, os.path.relpath(src, work))
                    os.makedirs(os.path.dirname(out), exist_ok=True)
                    shutil.copyfile(src, out)
    os.makedirs(dst, exist_ok=True)
    json.dump(info, open(os.path.join(dst, "log.json"), "w"), indent=1)


import threading  # noqa: E402
keep_case.lock = threading.Lock()

SELFCHECK = False  # --selfcheck: the checker's commands stand in for the product's (tests this tool where there is no GPU)


def one_case(case, seed, l2k11, poison, keep, shuf_dir):
    rs = np.random.RandomState(seed * 100003 + case)
    work = tempfile.mkdtemp(prefix="pfz%d_" % case, dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    log, verdict = [], None
    try:
        (k, subk, drl), refs, tax, qfiles = make_case(rs, work, l2k11)
        shuf = os.path.join(shuf_dir, "L%dK%d_%d.shuf" % (drl, k, subk))
        taxf = os.path.join(work, "tax.tsv")
        open(taxf, "w").write("".join(t + "\n" for t in tax))
        penv = dict(os.environ)
        if poison:
            penv["MK_POISON"] = poison
        pflags = [[], ["-p", "4"], ["-p", "16"], ["-p", "4", "--no-batch"], ["-p", "2", "--batch-text"]][rs.randint(0, 5)]
        # checker side
        steps_c = [("sk", [ORA, "-L", shuf, "-o", "c_sk"] + refs), ("grp", [REF, "set", "-g", taxf, "-o", "c_grp", "c_sk"]),
                   ("uq", [REF, "set", "-q", "-o", "c_uq", "c_grp"]), ("db", [REF, "set", "-i", "c_uq", "-o", "c_db", "c_grp"]),
                   ("qsk", [ORA, "-L", shuf, "-A", "-o", "c_qsk"] + qfiles)]
        steps_p = [("sk", [CLI, "dist", "-L", shuf, "-o", "p_sk"] + pflags + refs), ("grp", [CLI, "set", "-g", taxf, "-o", "p_grp", "p_sk"]),
                   ("uq", [CLI, "set", "-q", "-o", "p_uq", "p_grp"]), ("db", [CLI, "set", "-i", "p_uq", "-o", "p_db", "p_grp"]),
                   ("qsk", [CLI, "dist", "-L", shuf, "-A", "-o", "p_qsk"] + pflags[:2] + qfiles)]
        comp_p = [CLI, "composite"]
        if SELFCHECK:
            steps_p = [(n, [x.replace("c_", "p_") if x.startswith("c_") else x for x in c]) for n, c in steps_c]
            comp_p = [REF, "composite"]
        for (name, cc), (_, pc) in zip(steps_c, steps_p):
            rc_ = run(cc, work, log)
            rp_ = run(pc, work, log, env=penv)
            if rc_.returncode < 0:  # the reference died of a signal (division by zero ..): there is no behaviour to compare with
                verdict = "reference crashed in stage %s (signal %d), product rc %d" % (name, -rc_.returncode, rp_.returncode)
                break
            if (rc_.returncode != 0) != (rp_.returncode != 0):
                verdict = "stage %s: checker rc %d, product rc %d: %s" % (name, rc_.returncode, rp_.returncode, rp_.stderr.decode(errors="replace").strip()[-160:])
                break
            if rc_.returncode != 0:  # both refuse (an empty group, a crowded table ..): the case ends here, in agreement
                return {"case": case, "geom": [k, subk, drl], "ok": True, "ended": name}
            d = same_dir(os.path.join(work, "c_" + name), os.path.join(work, "p_" + name))
            if d:
                verdict = "stage %s: %s" % (name, d)
                break
        if verdict is None:
            rc_ = run([REF, "composite", "-r", "c_db", "-q", "c_qsk"], work, log)
            rp_ = run(comp_p + ["-r", "p_db", "-q", "p_qsk"], work, log, env=penv)
            if rc_.returncode < 0:
                verdict = "reference crashed in composite (signal %d), product rc %d" % (-rc_.returncode, rp_.returncode)
            elif rc_.returncode != rp_.returncode and (rc_.returncode == 0 or rp_.returncode == 0):
                verdict = "composite: checker rc %d, product rc %d: %s" % (rc_.returncode, rp_.returncode, rp_.stderr.decode(errors="replace").strip()[-160:])
            elif rc_.returncode == 0 and norm_composite(rc_.stdout.decode()) != norm_composite(rp_.stdout.decode()):
                verdict = "composite: text differs"
            elif rc_.returncode == 0:
                run([REF, "composite", "-r", "c_db", "-q", "c_qsk", "-b", "-o", "c_abv"], work, log)
                run(comp_p + ["-r", "p_db", "-q", "p_qsk", "-b", "-o", "p_abv"], work, log, env=penv)
                if os.path.isdir(os.path.join(work, "c_abv")) != os.path.isdir(os.path.join(work, "p_abv")):
                    verdict = "composite -b: one side wrote no directory"
                elif os.path.isdir(os.path.join(work, "c_abv")):
                    d = same_dir(os.path.join(work, "c_abv"), os.path.join(work, "p_abv"))
                    if d:
                        verdict = "composite -b: " + d
        if verdict and verdict.startswith("reference crashed"):
            return {"case": case, "geom": [k, subk, drl], "ok": True, "reference_crashed": verdict}
        if verdict:
            keep_case(keep, "case_%d_%d" % (seed, case), work, {"verdict": verdict, "geom": [k, subk, drl], "product_flags": pflags, "log": log})
        return {"case": case, "geom": [k, subk, drl], "ok": verdict is None, "what": verdict, "product_flags": pflags}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def golden_once(it, case, poison, keep, shuf_dir, first):
    """one run of a golden composite pipeline by the product; every stage against the first run's bytes, the end against the golden files"""
    import golden_cases as gc
    entry = json.load(open(os.path.join(gc.GOLDEN, "manifest.json")))["composite_cases"][case]
    work = tempfile.mkdtemp(prefix="gold%d_" % it, dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    log, verdict = [], None
    try:
        refs, qry = gc.build_composite_inputs(case, work)
        shuf = os.path.join(shuf_dir, entry["shuf"] + ".shuf")
        taxf = os.path.join(work, "tax.tsv")
        open(taxf, "w").write("".join(t + "\n" for t in entry["tax"]))
        penv = dict(os.environ)
        if poison:
            penv["MK_POISON"] = poison
        steps = [("sk", [CLI, "dist", "-p", "4", "-L", shuf, "-o", "sk"] + refs), ("grp", [CLI, "set", "-g", taxf, "-o", "grp", "sk"]),
                 ("uq", [CLI, "set", "-q", "-o", "uq", "grp"]), ("db", [CLI, "set", "-i", "uq", "-o", "db", "grp"]),
                 ("qsk", [CLI, "dist", "-p", "4", "-L", shuf, "-A", "-o", "qsk"] + qry)]
        for name, cmd in steps:
            r = run(cmd, work, log, env=penv)
            if r.returncode != 0:
                verdict = "stage %s: rc %d" % (name, r.returncode)
                break
            if first and os.path.isdir(os.path.join(first, name)):
                d = same_dir(os.path.join(first, name), os.path.join(work, name))
                if d:
                    verdict = "stage %s differs from the first run: %s" % (name, d)
                    break
        if verdict is None:
            exp = os.path.join(gc.GOLDEN, "expected", case)
            r = run([CLI, "composite", "-r", "db", "-q", "qsk"], work, log, env=penv)
            if r.returncode != 0 or norm_composite(r.stdout.decode()) != open(os.path.join(exp, "composite.tsv")).read().splitlines():
                verdict = "composite text (rc %d)" % r.returncode
            else:
                r = run([CLI, "composite", "-r", "db", "-q", "qsk", "-b", "-o", "abv"], work, log, env=penv)
                want = sorted(f for f in os.listdir(exp) if f.endswith(".abv"))
                if r.returncode != 0 or sorted(os.listdir(os.path.join(work, "abv"))) != want:
                    verdict = "composite -b listing (rc %d)" % r.returncode
                else:
                    for f in want:
                        if not filecmp.cmp(os.path.join(exp, f), os.path.join(work, "abv", f), shallow=False):
                            verdict = "composite -b: " + f
        if first and not os.path.isdir(os.path.join(first, "sk")) and verdict is None:
            for name, _ in steps:  # this run is the yardstick for the stages' bytes
                shutil.copytree(os.path.join(work, name), os.path.join(first, name))
        if verdict:
            keep_case(keep, "golden_%s_%d" % (case, it), work, {"verdict": verdict, "log": log})
        return {"it": it, "ok": verdict is None, "what": verdict}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--l2k11", type=float, default=0.15, help="share of the cases on the 16-component 537 M-slot geometry")
    ap.add_argument("--poison", default="", help="MK_POISON for the product's processes (e.g. 0xA5, 0x43)")
    ap.add_argument("--keep", default=os.path.join(ROOT, "gpurun_out", "fuzz_pipeline_fail"))
    ap.add_argument("--golden", default="", help="repeat this golden composite case instead of random cases")
    ap.add_argument("--times", type=int, default=100)
    ap.add_argument("--selfcheck", action="store_true", help="run the checker against itself (no GPU needed): tests this tool")
    a = ap.parse_args()
    global SELFCHECK
    SELFCHECK = a.selfcheck
    for p in (CLI, ORA) + (() if a.golden else (REF,)):
        if not os.path.exists(p):
            raise SystemExit("missing %s" % p)
    from metakssd_amd import capi
    os.makedirs(a.keep, exist_ok=True)
    shuf_dir = tempfile.mkdtemp(prefix="pfz_shuf_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    t0 = time.time()
    bad = []
    try:
        if a.golden:
            import golden_cases as gc
            name = json.load(open(os.path.join(gc.GOLDEN, "manifest.json")))["composite_cases"][a.golden]["shuf"]
            gc.make_shuf(name, os.path.join(shuf_dir, name + ".shuf"))
            first = os.path.join(shuf_dir, "first")
            os.makedirs(first)
            res = [golden_once(0, a.golden, a.poison, a.keep, shuf_dir, first)]  # the yardstick, alone
            with cf.ThreadPoolExecutor(max_workers=a.workers) as ex:
                res += list(ex.map(lambda i: golden_once(i, a.golden, a.poison, a.keep, shuf_dir, first), range(1, a.times)))
            bad = [r for r in res if not r["ok"]]
            print(json.dumps({"tool": "fuzz_pipeline", "golden": a.golden, "times": len(res), "workers": a.workers, "poison": a.poison or None,
                              "mismatches": len(bad), "what": [b["what"] for b in bad][:10], "seconds": round(time.time() - t0, 1)}))
        else:
            for k, subk, drl in GEOM_SMALL + [GEOM_L2K11]:
                capi.Shuf.generate(k, subk, drl, 700 + 16 * k + drl).write(os.path.join(shuf_dir, "L%dK%d_%d.shuf" % (drl, k, subk)))
            rs = np.random.RandomState(a.seed)
            l2 = rs.rand(a.cases) < a.l2k11
            with cf.ThreadPoolExecutor(max_workers=a.workers) as ex:
                res = list(ex.map(lambda i: one_case(i, a.seed, bool(l2[i]), a.poison, a.keep, shuf_dir), range(a.cases)))
            bad = [r for r in res if not r["ok"]]
            kinds = {}
            for r in bad:
                kind = r["what"].split(":")[0] + (": " + r["what"].split(": ")[-1][:90] if "rc" in r["what"] else "")
                kinds[kind] = kinds.get(kind, 0) + 1
            by_geom = {}
            for r in res:
                by_geom["L%dK%d" % (r["geom"][2], r["geom"][0])] = by_geom.get("L%dK%d" % (r["geom"][2], r["geom"][0]), 0) + 1
            print(json.dumps({"tool": "fuzz_pipeline", "cases": len(res), "seed": a.seed, "workers": a.workers, "poison": a.poison or None,
                              "by_geometry": by_geom, "ended_early_in_agreement": sum(1 for r in res if r.get("ended")),
                              "reference_crashed": sum(1 for r in res if r.get("reference_crashed")),
                              "mismatches": len(bad), "mismatch_kinds": kinds, "what": [(b["case"], b["what"]) for b in bad][:10],
                              "seconds": round(time.time() - t0, 1)}))
    finally:
        shutil.rmtree(shuf_dir, ignore_errors=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
