#!/usr/bin/env python3
"""tools/fuzz_parity.py -- randomized differential test of the HIP engine against the CPU oracle (GPU box only).

    python tools/fuzz_parity.py [--cases 200] [--seed 1]

Every case draws a .shuf geometry, a sketch flavour (-A counted / FASTA set / FASTA -u / FASTQ occurrence set), a row stride,
(half of the -A cases: the sketch merged from 2 / 3 / 5 shards through the multi-GPU ABI, by the gather or by key slices,)
a number of pushes and reads with random lengths, strands, lower case, N runs and odd bytes, and compares the engine's
sketch with the oracle's, bit for bit.  Exits non-zero on the first difference and prints how to reproduce it."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util_inputs as ui  # noqa: E402

GEOM = [(6, 3, 0), (7, 4, 1), (8, 4, 2), (8, 5, 2), (9, 5, 2), (9, 6, 3), (10, 6, 3), (10, 5, 2), (11, 6, 3), (7, 3, 0)]
GEOM_BIG = (11, 5, 2)  # 16 components, 537 M slots (4.3 GB oracle table, sparse bookkeeping in the engine): drawn rarely


def random_reads(rs, n, dense):
    pool = ui.rand_seq(rs, int(rs.choice([2000, 20000, 200000])))
    out = []
    lens = [[0, 1, 11, 12, 13, 21, 22, 23, 50, 100, 150, 151, 152, 250, 301], [150], [0, 22, 23, 60, 63, 95, 127, 150, 159],
            [100, 101, 150]][rs.randint(0, 4)]  # sometimes fixed-length or short reads: the narrow-row (one-pass staging) kernels
    for _ in range(n):
        L = int(rs.choice(lens))
        if L > len(pool):
            L = len(pool)
        a = rs.randint(0, len(pool) - L + 1)
        s = pool[a:a + L]
        r = rs.rand()
        if r < 0.4:
            s = ui.revcomp(s)
        if rs.rand() < 0.2:
            s = s.lower()
        if L > 4 and rs.rand() < 0.25:
            j = rs.randint(0, L - 2)
            s = s[:j] + bytes(rs.choice([78, 110, 45, 42, 85, 0x80 if False else 88], size=rs.randint(1, 3)).astype(np.uint8)) + s[j + 2:]
            s = s[:L]
        out.append(s)
    return out


def _hip():
    return ctypes.CDLL("libamdhip64.so")


def _dmalloc(hip, held, nbytes):
    q = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(max(16, nbytes))) == 0
    held.append(q)
    return q.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--start", type=int, default=0, help="first case (every case draws from its own stream: --start N --cases 1 replays case N)")
    ap.add_argument("--trace", action="store_true", help="print every case's number before it runs and its description behind it")
    ap.add_argument("--big", action="store_true", help="also draw the 16-component L2K11 geometry (slow oracle)")
    ap.add_argument("--big-rate", type=float, default=0.08, help="share of the cases that draw it (with --big)")
    ap.add_argument("--sparse", type=int, default=-1, help="MK_OPT_SPARSE: -1 by table size, 0 off, 1 on")
    ap.add_argument("--front-bits", type=int, default=None, help="MK_OPT_FRONT_BITS: front table of 2^n slots (with --sparse 0)")
    ap.add_argument("--split-cus", type=int, default=0, help="MK_OPT_SPLIT_CUS on every engine (32, 64, ..): the scan kernel on a queue of its own")
    ap.add_argument("--tail-rate", type=float, default=0.0,
                    help="with --split-cus: this share of the sketches begin with MK_BEGIN_NOTHING_FOLLOWS (their tail on the engine's unmasked queue)")
    a = ap.parse_args()
    from metakssd_amd import capi
    from oracle_binding import Oracle
    if a.tail_rate > 0:  # every Engine.begin of the run draws: the engine goes to its unmasked queue and back between sketches of all kinds
        _begin0 = capi.Engine.begin
        trs = np.random.RandomState(a.seed ^ 0x5A5A)

        def _begin(self, mode=capi.MK_MODE_KOC):
            return _begin0(self, mode | (capi.MK_BEGIN_NOTHING_FOLLOWS if trs.rand() < a.tail_rate else 0))
        capi.Engine.begin = _begin
    engines, oracles, shufs = {}, {}, {}
    bad = 0
    nonempty = total_ids = crowded = 0
    for case in range(a.start, a.start + a.cases):
        if a.trace:
            print("case", case, flush=True)
        rs = np.random.RandomState(a.seed * 100003 + case)
        k, subk, drl = GEOM_BIG if (a.big and rs.rand() < a.big_rate) else GEOM[rs.randint(0, len(GEOM))]
        key = (k, subk, drl)
        if key not in shufs:
            shufs[key] = capi.Shuf.generate(k, subk, drl, 1000 + k * 100 + subk * 10 + drl)
            engines[key] = capi.Engine(shufs[key], 0, sparse=a.sparse, front_bits=a.front_bits)
            if a.split_cus:
                engines[key].set_option(capi.MK_OPT_SPLIT_CUS, a.split_cus)
            oracles[key] = Oracle(shufs[key].c.id, k, subk, drl, shufs[key].table)
        eng, ora, P = engines[key], oracles[key], shufs[key].params()
        dense = P.dim_end - P.dim_start >= 16 ** subk  # accept-everything tables crowd quickly
        flavour = ["koc", "set", "uniq", "occ"][rs.randint(0, 4)]
        budget = int(P.hashlimit * (0.5 if flavour != "occ" else 0.8))
        per_read = 100 * (1.0 if dense else max(1.0 / 16 ** drl, 1e-4))
        nreads = int(min(rs.choice([1, 7, 64, 65, 300, 2000]), max(1, budget / max(per_read, 1e-9) / 2)))
        desc = "case %d seed %d: k=%d subk=%d drlevel=%d %s nreads=%d" % (case, a.seed, k, subk, drl, flavour, nreads)
        seqs = random_reads(rs, nreads, dense)
        if flavour in ("koc",):
            need = max([len(x) for x in seqs] + [0]) + 1
            stride = int(rs.choice([x for x in (64, 96, 128, 152, 160, 164, 176, 304, 308, 320, 512, 4096) if x >= need]))
            rows = ui.rows_from_seqs(seqs, stride)
            rc, want = ora.koc_from_rows(rows, stride)
            desc += " stride=%d" % stride
            tuned = bool(capi.lib.mk_params_packed_ok(ctypes.byref(P)))
            packed = tuned and need <= 153 and rs.rand() < 0.5   # 64-byte packed rows (mk_scan_packed_kernel)
            desc += " packed=%d" % packed
            merge = [None, None, "gather", "slices"][rs.randint(0, 4)] if rc == 0 else None
            if merge:
                # SURVEY 8e through the C ABI, on this one engine shard after shard: contiguous read ranges with global ordinals, every
                # shard's list exported (whole, or cut by key % G); then one sketch that imports them all (gather), or G sketches that each
                # fold one key slice, whose reduced lists are adopted as the last sketch's key list (slices)
                G = int(rs.choice([2, 3, 5]))
                n = len(seqs)
                cuts = [n * g // G for g in range(G + 1)]
                desc += " merge=%s G=%d" % (merge, G)
                hip = _hip()
                held, shards = [], []
                try:
                    for g in range(G):
                        eng.begin(capi.MK_MODE_KOC)
                        eng.push_reads(rows[cuts[g] * stride:cuts[g + 1] * stride], stride, cuts[g])
                        d = eng.partial_count()
                        bk, bc, bo = _dmalloc(hip, held, 8 * d), _dmalloc(hip, held, 4 * d), _dmalloc(hip, held, 8 * d)
                        if merge == "gather":
                            assert eng.partial_export(bk, bc, bo, d) == d
                            parts = [d]
                        else:
                            got_n, parts = eng.partial_export_split(G, bk, bc, bo, d)
                            assert got_n == d and sum(parts) == d
                        shards.append((bk, bc, bo, parts))
                        eng.finish()  # (the shard's own sketch: spent)
                    if merge == "gather":
                        eng.begin(capi.MK_MODE_KOC)
                        for bk, bc, bo, parts in shards:
                            if parts[0]:
                                eng.partial_import(bk, bc, bo, parts[0])
                    else:
                        slices = []
                        for g in range(G):
                            eng.begin(capi.MK_MODE_KOC)
                            for bk, bc, bo, parts in shards:
                                off = sum(parts[:g])
                                if parts[g]:
                                    eng.partial_import(bk + 8 * off, bc + 4 * off, bo + 8 * off, parts[g])
                            r = eng.partial_count()
                            lk, lc, lo = eng.partial_list_reserve(r)
                            ck, cc, co = _dmalloc(hip, held, 8 * r), _dmalloc(hip, held, 4 * r), _dmalloc(hip, held, 8 * r)
                            eng.sync()
                            for dst, src, w in ((ck, lk, 8), (cc, lc, 4), (co, lo, 8)):
                                assert hip.hipMemcpy(ctypes.c_void_p(dst), ctypes.c_void_p(src), ctypes.c_size_t(w * r), 3) == 0
                            assert hip.hipDeviceSynchronize() == 0  # (a device-to-device hipMemcpy does not wait for its copy; the finish below hands the list back)
                            slices.append((ck, cc, co, r))
                            eng.finish()
                        eng.begin(capi.MK_MODE_KOC)
                        at = 0
                        for ck, cc, co, r in slices:
                            eng.partial_list_adopt(ck, cc, co, r, at)
                            at += r
                        eng.partial_list_commit(at)
                    # (finished below, the buffers live until then)
                    try:
                        got = eng.finish()
                        grc = 0
                    except capi.CrowdedError:
                        grc, got = -2, None
                finally:
                    for q in held:
                        hip.hipFree(q)
                n_ids = sum(len(w[0]) for w in want)
                ok = grc == 0 and len(got) == len(want) and all(np.array_equal(g_[0], w[0]) and np.array_equal(g_[1], w[1]) for g_, w in zip(got, want))
                nonempty += n_ids > 0
                total_ids += n_ids
                if not ok:
                    bad += 1
                    print("MISMATCH", desc, "oracle rc", rc, "engine rc", grc)
                    break
                continue
            eng.begin(capi.MK_MODE_KOC)
            pushes = int(rs.choice([1, 2, 5]))
            n = len(seqs)
            per = (n + pushes - 1) // pushes
            prows = capi.pack_rows_host(rows, stride) if packed else None
            for s0 in range(0, n, per):
                if packed and rs.rand() < 0.8:  # (a sketch may mix the two row formats)
                    eng.push_reads(prows[s0 * 64:(s0 + per) * 64], 64 | capi.MK_ROWS_PACKED, s0)
                else:
                    eng.push_reads(rows[s0 * stride:(s0 + per) * stride], stride, s0)
        elif flavour == "occ":
            M, Q = int(rs.choice([1, 2, 3, 7])), int(rs.choice([0, 40, 54]))
            quals = ui.random_quals(rs, seqs)
            data = ui.fastq_bytes(seqs, quals=quals, final_newline=bool(rs.rand() < 0.8))
            rc, want = ora.co_from_fastq(data, Q=Q, M=M)
            stride = int(rs.choice([304, 512]))
            rows, nrows, nrec, used, frc = capi.fastq_frame_q(data, stride, P.TL, qmin=Q)
            assert frc == 0 and used == len(data), desc
            eng.begin_occ(M)
            eng.push_reads(rows, stride, 0)
            desc += " M=%d Q=%d" % (M, Q)
        else:
            fa = ui.fasta_bytes([s for s in seqs if len(s) > 0] or [b"ACGT"], width=int(rs.choice([60, 70, 80, 1000])))
            if rs.rand() < 0.5:  # rougher text: CRLF, blank lines, '>' inside a sequence line, a header without sequence
                lines = fa.split(b"\n")
                for _ in range(int(rs.randint(1, 6))):
                    j = int(rs.randint(0, len(lines)))
                    what = rs.randint(0, 5)
                    if what == 0:
                        lines.insert(j, b"")
                    elif what == 1 and lines[j] and not lines[j].startswith(b">"):
                        cut = int(rs.randint(0, len(lines[j]) + 1))
                        lines[j] = lines[j][:cut] + b">" + lines[j][cut:]
                    elif what == 2:
                        lines.insert(j, b">only a header " + bytes(rs.randint(32, 127, size=int(rs.randint(0, 90))).astype(np.uint8)).replace(b"\n", b" "))
                    elif what == 3:
                        lines[j] = lines[j] + b"\r"
                    else:
                        lines[j] = lines[j].replace(b"A", b"a", 3)
                fa = b"\n".join(lines)
                if rs.rand() < 0.3 and fa.endswith(b"\n"):
                    fa = fa[:-1]
            if rs.rand() < 0.3:  # a BATCH of files (mk_sketch_batch_begin / _end): this text cut at header lines into 1..6 files
                lines = fa.split(b"\n")
                heads = [i for i, ln in enumerate(lines) if ln.startswith(b">")][1:]
                cuts = sorted(set(int(x) for x in rs.choice(heads, size=min(len(heads), int(rs.randint(0, 6))), replace=False))) if heads else []
                parts, prev = [], 0
                for cpos in cuts + [len(lines)]:
                    parts.append(b"\n".join(lines[prev:cpos]) + (b"\n" if cpos < len(lines) else b""))
                    prev = cpos
                parts = [x for x in parts if x] or [fa]
                tb = int(rs.choice([0, 0, 9, 12]))
                eng.set_option(capi.MK_OPT_BATCH_TAB_BITS, tb)
                mode = capi.MK_MODE_UNIQ_SET if flavour == "uniq" else capi.MK_MODE_SET
                how = int(rs.randint(0, 3)) if capi.lib.mk_params_packed_ok(ctypes.byref(P)) else 0
                if a.trace:
                    print(desc, "batch how=%d tb=%d parts=%s" % (how, tb, [len(x) for x in parts]), flush=True)
                if how == 0:    # the texts; the device walks them
                    eng.batch_begin(parts, mode, one_buffer=bool(rs.rand() < 0.5))
                    live = list(range(len(parts)))
                else:           # the host walks them (mk_fasta_pack_rows): narrow or wide rows, read in place or copied
                    fmt = capi.MK_ROWS_PACKED if how == 1 else capi.MK_ROWS_WIDE
                    packed = [capi.fasta_pack_rows(x, P.TL, fmt) for x in parts]
                    live = [i for i, (_, prc_) in enumerate(packed) if prc_ == 0]   # (a text ending inside a header is refused by the packer: checked below)
                    if live:
                        eng.batch_begin_rows([packed[i][0] for i in live], mode, pinned=bool(rs.rand() < 0.6), fmt=fmt, gap=int(rs.choice([0, 64, 6400])))
                res_live = eng.batch_end() if (how == 0 or live) else []
                res = [None] * len(parts)
                for i, r_ in zip(live, res_live):
                    res[i] = r_
                if how:
                    for i, (_, prc_) in enumerate(packed):
                        if prc_ != 0:
                            res[i] = (prc_, 0, [])
                eng.set_option(capi.MK_OPT_BATCH_TAB_BITS, 0)
                desc += " batch of %d files tb=%d %s" % (len(parts), tb, ["texts", "narrow rows", "wide rows"][how])
                okb = True
                for part, (st, alone, comps) in zip(parts, res):
                    prc, pw = ora.co_from_fasta(part, uniq=flavour == "uniq")
                    if prc == -5:
                        okb = okb and st == capi.MK_ERR_FORMAT
                    elif prc != 0:
                        okb = okb and st != 0
                    else:
                        okb = okb and st == 0 and len(comps) == len(pw) and all(np.array_equal(g, w[0]) for g, w in zip(comps, pw))
                        total_ids += sum(len(w[0]) for w in pw)
                nonempty += 1
                if not okb:
                    bad += 1
                    print("MISMATCH", desc)
                    break
                continue
            rc, want = ora.co_from_fasta(fa, uniq=flavour == "uniq")
            stride = int(rs.choice([64, 256, 512, 4096]))
            if stride < 2 * P.TL + 4:
                stride = 256
            eng.begin(capi.MK_MODE_UNIQ_SET if flavour == "uniq" else capi.MK_MODE_SET)
            if rs.rand() < 0.5:  # the text itself to the device (mk_sketch_push_stream), whole or in pieces cut anywhere
                piece = int(rs.choice([0, 0, 37, 1000, 70001])) or None
                desc += " device stream piece=%s" % piece
                eng.push_stream(fa, piece=piece)
                if rc == -5:  # input ends inside a '>' line: the reference aborts there; the engine reports it at finish
                    try:
                        eng.finish()
                        ok_format = False
                    except capi.MkError as ex:
                        ok_format = ex.code == capi.MK_ERR_FORMAT
                    if not ok_format:
                        bad += 1
                        print("MISMATCH", desc, "the oracle refuses the text (ends inside a header), the engine did not")
                    crowded += 1
                    continue
            else:
                try:
                    rows = capi.fasta_windows(fa, P.TL, stride, chunk=int(rs.choice([0, 100, 5000])) or None)
                    eng.push_reads(rows, stride, 0)
                except capi.MkError as ex:  # input ends inside a '>' line: the reference aborts there, the oracle says KO_ERR_CONTRACT
                    if ex.code != capi.MK_ERR_FORMAT or rc != -5:
                        raise
                    eng.finish()
                    crowded += 1
                    continue
        try:
            got = eng.finish()
            grc = 0
        except capi.CrowdedError:
            grc, got = -2, None
        if rc != 0 or grc != 0:
            ok = (rc != 0) == (grc != 0)
            crowded += 1
        else:
            n_ids = sum(len(w[0]) for w in want)
            nonempty += n_ids > 0
            total_ids += n_ids
            ok = len(got) == len(want) and all(np.array_equal(g[0], w[0]) and (w[1] is None or np.array_equal(g[1], w[1])) for g, w in zip(got, want))
        if not ok:
            bad += 1
            print("MISMATCH", desc, "oracle rc", rc, "engine rc", grc)
            break
    print("%d cases, %d mismatches; %d with a non-empty sketch (%d ids compared), %d crowded on both sides" %
          (case + 1, bad, nonempty, total_ids, crowded))
    for e in engines.values():
        e.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
