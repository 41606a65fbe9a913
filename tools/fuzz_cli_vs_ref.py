#!/usr/bin/env python3
"""tools/fuzz_cli_vs_ref.py -- end to end: the product CLI against the COMPILED REFERENCE (oracle/_ref/metakssd) on random
single-file inputs, sketch directories compared byte for byte (cofiles.stat field-wise).  GPU box with oracle/_ref present.

    python tools/fuzz_cli_vs_ref.py [--cases 100] [--seed 1]
One input file per run (the reference permutes multi-file inputs with a time seed)."""
import argparse
import filecmp
import gzip
import os
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util_inputs as ui  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "metakssd")
CLI = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
GEOM = [(6, 3, 0), (7, 4, 1), (8, 4, 2), (9, 5, 2), (9, 6, 3), (10, 6, 3), (11, 6, 3)]


def stat_fields(path):
    b = open(path, "rb").read()
    n = struct.unpack_from("<i", b, 20)[0]
    names = [b[32 + 4 * n + 256 * i:32 + 4 * n + 256 * (i + 1)].split(b"\0", 1)[0] for i in range(n)]
    return b[:5], b[8:32 + 4 * n], names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from metakssd_amd import capi
    work = tempfile.mkdtemp(prefix="clifuzz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    shufs = {}
    bad = aborted = 0
    for case in range(a.cases):
        rs = np.random.RandomState(a.seed * 9973 + case)
        k, subk, drl = GEOM[rs.randint(0, len(GEOM))]
        if (k, subk, drl) not in shufs:
            p = os.path.join(work, "L%dK%d_%d.shuf" % (drl, k, subk))
            capi.Shuf.generate(k, subk, drl, 500 + k).write(p)
            shufs[(k, subk, drl)] = p
        dense = subk - drl >= 3 and 16 ** (subk - drl) <= 4096
        kind = ["fq_A", "fq_set", "fa", "fa_u"][rs.randint(0, 4)]
        nreads = int(rs.choice([1, 5, 60, 400] if dense else [1, 5, 60, 400, 3000]))
        pool = ui.rand_seq(rs, int(rs.choice([3000, 30000, 300000])))
        seqs = []
        for _ in range(nreads):
            L = int(rs.choice([0, 1, 14, 22, 23, 100, 150, 151, 250]))
            s0 = pool[rs.randint(0, len(pool) - L + 1):][:L]
            if rs.rand() < 0.4:
                s0 = ui.revcomp(s0)
            if rs.rand() < 0.15:
                s0 = s0.lower()
            if L > 3 and rs.rand() < 0.2:
                j = rs.randint(0, L - 1)
                s0 = s0[:j] + b"N" + s0[j + 1:]
            seqs.append(s0)
        flags = []
        if kind.startswith("fq"):
            quals = ui.random_quals(rs, seqs)
            data = ui.fastq_bytes(seqs, crlf=bool(rs.rand() < 0.1), final_newline=bool(rs.rand() < 0.85),
                                  drop_last_qual=bool(rs.rand() < 0.05), quals=quals)
            name = "in.fq" if rs.rand() < 0.7 else "in.fastq"
            if kind == "fq_A":
                flags = ["-A"]
            else:
                flags = ["-n", str(int(rs.choice([1, 2, 3, 7, 9]))), "-Q", str(int(rs.choice([0, 36, 54, 74])))]
        else:
            data = ui.fasta_bytes([s0 for s0 in seqs if s0] or [b"ACGTACGT"], width=int(rs.choice([60, 70, 80, 500])))
            if rs.rand() < 0.2:
                data = data.replace(b"\n", b"\r\n")
            name = ["in.fa", "in.fasta", "in.fna"][rs.randint(0, 3)]
            flags = ["-u"] if kind == "fa_u" else []
        path = os.path.join(work, name)
        if rs.rand() < 0.25:
            path += ".gz"
            with gzip.GzipFile(path, "wb", mtime=0) as f:
                f.write(data)
        else:
            open(path, "wb").write(data)
        o_ref, o_cli = os.path.join(work, "ref"), os.path.join(work, "cli")
        shutil.rmtree(o_ref, ignore_errors=True)
        shutil.rmtree(o_cli, ignore_errors=True)
        r1 = subprocess.run([REF, "dist", "-L", shufs[(k, subk, drl)]] + flags + ["-p", "1", "-o", o_ref, path], cwd=work,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        r2 = subprocess.run([CLI, "dist", "-L", shufs[(k, subk, drl)]] + flags + ["-o", o_cli, "--quiet", path], cwd=work,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        desc = "case %d seed %d: k=%d subk=%d drl=%d %s %s %s nreads=%d" % (case, a.seed, k, subk, drl, kind, " ".join(flags), os.path.basename(path), nreads)
        ref_done = os.path.exists(os.path.join(o_ref, "cofiles.stat"))
        if not ref_done:  # the reference gave up (too crowd, header at EOF ...): the product must fail too
            aborted += 1
            ok = r2.returncode != 0
        else:
            ok = r2.returncode == 0
            if ok:
                names = sorted(f for f in os.listdir(o_ref) if f.startswith("combco"))
                ok = names == sorted(f for f in os.listdir(o_cli) if f.startswith("combco"))
                ok = ok and all(filecmp.cmp(os.path.join(o_ref, f), os.path.join(o_cli, f), shallow=False) for f in names)
                ok = ok and stat_fields(os.path.join(o_ref, "cofiles.stat")) == stat_fields(os.path.join(o_cli, "cofiles.stat"))
        os.remove(path)
        if not ok:
            bad += 1
            print("MISMATCH", desc, "ref rc", r1.returncode, "cli rc", r2.returncode, r2.stderr.decode(errors="replace")[-200:])
            keep = os.path.join(ROOT, "gpurun_out", "clifuzz_case_%d_%d" % (a.seed, case))
            os.makedirs(keep, exist_ok=True)
            open(os.path.join(keep, os.path.basename(path)), "wb").write(data if not path.endswith(".gz") else gzip.compress(data))
            break
    print("%d cases, %d mismatches, %d where the reference gave up (and the product failed too)" % (case + 1, bad, aborted))
    shutil.rmtree(work, ignore_errors=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
