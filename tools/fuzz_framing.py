#!/usr/bin/env python3
"""tools/fuzz_framing.py -- CPU-only: the product's FASTQ framers (serial and threaded) against the oracle's restatements of
the reference's two readers on random, partly malformed FASTQ-like text (blank lines, missing lines, CR, no final newline).
    python tools/fuzz_framing.py [--cases 2000] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util_inputs as ui  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from metakssd_amd import capi
    from oracle_binding import Oracle
    sh = capi.Shuf.generate(7, 4, 1, 7)
    ora = Oracle(sh.c.id, 7, 4, 1, sh.table)
    bad = 0
    for case in range(a.cases):
        rs = np.random.RandomState(a.seed * 65537 + case)
        nrec = int(rs.choice([0, 1, 2, 3, 10, 60]))
        lines = []
        for r in range(nrec):
            s = ui.rand_seq(rs, int(rs.choice([0, 5, 14, 15, 40, 150])))
            if rs.rand() < 0.2:
                s = s[: len(s) // 2] + b"N" + s[len(s) // 2:]
            q = bytes(rs.choice(np.frombuffer(b"#5I", np.uint8), size=len(s)).astype(np.uint8))
            rec = [b"@r%d" % r, s, b"+", q]
            if rs.rand() < 0.08:
                rec.insert(int(rs.randint(0, 5)), b"")            # a stray blank line shifts everything behind it
            if rs.rand() < 0.05:
                del rec[int(rs.randint(0, len(rec)))]             # a missing line
            lines += rec
        nl = b"\r\n" if rs.rand() < 0.1 else b"\n"
        data = nl.join(lines) + (nl if lines and rs.rand() < 0.8 else b"")
        desc = "case %d seed %d (%d bytes)" % (case, a.seed, len(data))
        # -A reader
        rc, want = ora.koc_from_fastq(data)
        rows, n, used, frc = capi.fastq_frame(data, 304)
        ok = frc == 0 and rc == 0 and n == ora.last_nreads
        if ok:
            rc2, got = ora.koc_from_rows(rows, 304) if n else (0, [(np.zeros(0, np.uint32), np.zeros(0, np.uint16))])
            ok = rc2 == 0 and np.array_equal(got[0][0], want[0][0]) and np.array_equal(got[0][1], want[0][1])
        # fastq2co reader (skip inputs whose first fgets round fails: undefined in the reference)
        if ok and len(lines) >= 4:
            for Q, M in ((0, 1), (54, 2)):
                rc, want = ora.co_from_fastq(data, Q=Q, M=M)
                rows, n, nrecs, used, frc = capi.fastq_frame_q(data, 304, 14, qmin=Q)
                if rc != 0 or frc != 0:
                    ok = False
                    break
                rc2, res = ora.koc_from_rows(rows, 304) if n else (0, [(np.zeros(0, np.uint32), np.zeros(0, np.uint16))])
                ok = rc2 == 0 and np.array_equal(res[0][0][res[0][1] >= M], want[0][0])
                if not ok:
                    desc += " occ Q=%d M=%d" % (Q, M)
                    break
        if not ok:
            bad += 1
            print("MISMATCH", desc)
            open("/tmp/fuzz_framing_case.fq", "wb").write(data)
            break
    print("%d cases, %d mismatches" % (case + 1, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
