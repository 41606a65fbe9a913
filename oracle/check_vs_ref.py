#!/usr/bin/env python3
"""oracle/check_vs_ref.py -- TEST INFRASTRUCTURE ONLY.

Pins the oracle restatement (oracle/kssd_oracle_cli) against the REAL reference
(oracle/_ref/metakssd, compiled from /root/reference by `make -C oracle ref`):
both run `dist -L <.shuf> [-A] [-u] -p 1 -o <dir> <inputs>` on the same generated inputs and the
payload files (combco.N, combco.N.a, combco.index.N) must be byte-identical; cofiles.stat is
compared field-wise (the reference leaves 3 padding bytes uninitialised).

Runs only where oracle/_ref/metakssd exists (this container).  Usage:
    python oracle/check_vs_ref.py [--keep DIR] [--big]
"""
import argparse
import ctypes
import filecmp
import os
import random
import shutil
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.path.join(HERE, "_ref", "metakssd")
ORA = os.path.join(HERE, "kssd_oracle_cli")


def host_lib():
    """the product's host-side C helpers (.shuf generator, synthetic FASTQ writer) -- input generators only"""
    for cand in (os.path.join(ROOT, "metakssd_amd", "lib", "libmetakssd_hip.so"), "/tmp/libmkhost.so"):
        if os.path.exists(cand):
            return ctypes.CDLL(cand)
    raise SystemExit("build the product library first (python -c 'import __graft_entry__ as g; g.build()')")


class Shuf(ctypes.Structure):
    _fields_ = [("id", ctypes.c_int32), ("k", ctypes.c_int32), ("subk", ctypes.c_int32), ("drlevel", ctypes.c_int32),
                ("table", ctypes.POINTER(ctypes.c_int32)), ("len", ctypes.c_uint64)]


def gen_shuf(lib, path, k, subk, drl, seed):
    s = Shuf()
    rc = lib.mk_shuf_generate(k, subk, drl, ctypes.c_uint64(seed), ctypes.byref(s))
    assert rc == 0, rc
    assert lib.mk_shuf_write(ctypes.byref(s), path.encode()) == 0
    lib.mk_shuf_free(ctypes.byref(s))


def parse_stat(path):
    b = open(path, "rb").read()
    shuf_id, koc = struct.unpack_from("<IB", b, 0)
    kmerlen, dim_rd_len, comp_num, infile_num, all_ctx = struct.unpack_from("<iiiiQ", b, 8)
    cts = struct.unpack_from("<%dI" % infile_num, b, 32)
    names = []
    off = 32 + 4 * infile_num
    for i in range(infile_num):
        raw = b[off + 256 * i: off + 256 * (i + 1)]
        names.append(raw.split(b"\0", 1)[0].decode())
    assert len(b) == off + 256 * infile_num
    return dict(shuf_id=shuf_id, koc=koc, kmerlen=kmerlen, dim_rd_len=dim_rd_len, comp_num=comp_num,
                infile_num=infile_num, all_ctx=all_ctx, cts=list(cts), names=names)


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")


def compare_dirs(a, b, single_input=True):
    """payload byte-equality + stat field equality; returns list of problems"""
    bad = []
    sa, sb = parse_stat(os.path.join(a, "cofiles.stat")), parse_stat(os.path.join(b, "cofiles.stat"))
    if sa != sb:
        bad.append("cofiles.stat fields differ: %r vs %r" % (sa, sb))
    names = sorted(f for f in os.listdir(a) if f.startswith("combco"))
    namesb = sorted(f for f in os.listdir(b) if f.startswith("combco"))
    if names != namesb:
        bad.append("file sets differ: %r vs %r" % (names, namesb))
    for f in names:
        if f in namesb and not filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False):
            bad.append("%s differs" % f)
    return bad


RC = str.maketrans("ACGTacgt", "TGCAtgca")


def revcomp(s):
    return s.translate(RC)[::-1]


def write_fq(path, seqs, crlf=False, final_newline=True, drop_last_qual=False, quals=None):
    nl = "\r\n" if crlf else "\n"
    out = []
    for i, s in enumerate(seqs):
        rec = ["@r%d" % i, s, "+", quals[i] if quals else "I" * len(s)]
        if drop_last_qual and i == len(seqs) - 1:
            rec = rec[:3]
        out.append(nl.join(rec) + nl)
    txt = "".join(out)
    if not final_newline:
        txt = txt[:-len(nl)]
    open(path, "w", newline="").write(txt)


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep", default=None)
    ap.add_argument("--big", action="store_true", help="also run the 1.5 M-read K9 collision case")
    ap.add_argument("--search", action="store_true", help="also stage II + `dist -r` (the reference writes 32 GiB per database: minutes each)")
    ap.add_argument("--dense", action="store_true", help="with --search: the oracle writes its own 32 GiB index too and the two are compared")
    args = ap.parse_args()
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/metakssd missing: run `make -C oracle ref` where /root/reference exists")
    subprocess.check_call(["make", "-s", "-C", HERE])
    lib = host_lib()
    lib.mk_shuf_generate.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_uint64, ctypes.POINTER(Shuf)]
    lib.mk_synth_fastq_write.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
    work = args.keep or tempfile.mkdtemp(prefix="kssd_chk_")
    os.makedirs(work, exist_ok=True)
    rng = random.Random(20260101)
    failures = 0

    shufs = {}
    # L0K6 (subk 3, level 0): dim_end = max(16^3,4096) = whole space => EVERY k-mer accepted, table 131071 slots:
    #   a few hundred reads give load ~0.5 (collision order), a few thousand overflow hashlimit (abort path).
    # L1K7 (subk 4, level 1): 1/16 of k-mers accepted, same table size.
    for name, (k, subk, drl, seed) in {"L3K11": (11, 6, 3, 11), "L3K9": (9, 6, 3, 9), "L3K10": (10, 6, 3, 10),
                                       "L2K11": (11, 5, 2, 211), "L0K6": (6, 3, 0, 6), "L1K7": (7, 4, 1, 7)}.items():
        p = os.path.join(work, name + ".shuf")
        if not os.path.exists(p):
            gen_shuf(lib, p, k, subk, drl, seed)
        shufs[name] = p

    # L0K6 variant whose table maps inner substring 0 to 0, so poly-A / poly-T reads produce key 0
    import array
    zpath = os.path.join(work, "L0K6z.shuf")
    raw = open(shufs["L0K6"], "rb").read()
    tab = array.array("i", raw[16:])
    j = tab.index(0)
    tab[j], tab[0] = tab[0], 0
    open(zpath, "wb").write(raw[:16] + tab.tobytes())
    shufs["L0K6z"] = zpath

    def case(label, shuf, inputs, flags, expect_abort=False):
        nonlocal failures
        o_ref, o_ora = os.path.join(work, label + ".ref"), os.path.join(work, label + ".ora")
        for d in (o_ref, o_ora):
            shutil.rmtree(d, ignore_errors=True)
        rc1, out1, err1 = run([REF, "dist", "-L", shufs[shuf]] + flags + ["-p", "1", "-o", o_ref] + inputs, work)
        rc2, out2, err2 = run([ORA, "-L", shufs[shuf]] + flags + ["-o", o_ora] + inputs, work)
        # the reference aborts through err(errno, "...too crowd...") (iseq2comem.c:708-709); errno is 0 there,
        # so its exit status is 0: the abort is recognised by the message, and by cofiles.stat never being written
        ref_abort = ("too crowd" in err1 or "can not find seqences head" in err1) and not os.path.exists(os.path.join(o_ref, "cofiles.stat"))
        if expect_abort or ref_abort:
            ok = expect_abort and ref_abort and rc2 != 0
            print("%s %-28s both abort: ref abort=%s ora rc=%d" % ("ok  " if ok else "FAIL", label, ref_abort, rc2))
            failures += 0 if ok else 1
            return
        if rc1 != 0 or rc2 != 0:
            print("FAIL %-28s ref rc=%d ora rc=%d\n%s\n%s" % (label, rc1, rc2, err1[-300:], err2[-300:]))
            failures += 1
            return
        bad = compare_dirs(o_ref, o_ora)
        st = parse_stat(os.path.join(o_ref, "cofiles.stat"))
        if bad:
            failures += 1
            print("FAIL %-28s %s" % (label, "; ".join(bad)))
        else:
            print("ok   %-28s distinct=%d comps=%d" % (label, st["all_ctx"], st["comp_num"]))

    # ---- FASTQ -A cases -------------------------------------------------------------------
    fq = os.path.join(work, "syn100k.fq")
    lib.mk_synth_fastq_write(fq.encode(), 1, 0, 100000, 150)
    case("L3K11_syn100k", "L3K11", [fq], ["-A"])          # BASELINE config 1
    case("L3K9_syn100k", "L3K9", [fq], ["-A"])            # hashsize 131071, load ~0.02

    # dense-collision case: K9 table (131071 slots), reads drawn from a small genome pool => many repeats
    pool = rand_seq(rng, 200000)
    seqs = []
    for i in range(60000):
        a = rng.randrange(0, len(pool) - 150)
        s = pool[a:a + 150]
        if rng.random() < 0.5:
            s = revcomp(s)
        if rng.random() < 0.05:
            j = rng.randrange(150)
            s = s[:j] + "N" + s[j + 1:]
        seqs.append(s)
    fq2 = os.path.join(work, "pool60k.fq")
    write_fq(fq2, seqs)
    case("L3K9_pool60k", "L3K9", [fq2], ["-A"])
    case("L3K11_pool60k", "L3K11", [fq2], ["-A"])

    # ragged lengths, lower case, N runs, CRLF, truncated last record, no final newline
    seqs = []
    for i in range(20000):
        L = rng.choice([0, 1, 21, 22, 23, 50, 100, 150, 151, 250, 300])
        s = rand_seq(rng, L)
        if rng.random() < 0.3:
            s = s.lower()
        if L > 30 and rng.random() < 0.3:
            j = rng.randrange(L - 5)
            s = s[:j] + "NNN" + s[j + 3:]
        seqs.append(s)
    fq3 = os.path.join(work, "ragged.fq")
    write_fq(fq3, seqs)
    case("L3K9_ragged", "L3K9", [fq3], ["-A"])
    fq4 = os.path.join(work, "ragged_crlf.fq")
    write_fq(fq4, seqs[:5000], crlf=True)
    case("L3K9_ragged_crlf", "L3K9", [fq4], ["-A"])
    fq5 = os.path.join(work, "trunc.fq")
    write_fq(fq5, seqs[:3001], drop_last_qual=True)
    case("L3K9_trunc_last", "L3K9", [fq5], ["-A"])
    fq6 = os.path.join(work, "nonl.fq")
    write_fq(fq6, seqs[:3000], final_newline=False)
    case("L3K9_no_final_nl", "L3K9", [fq6], ["-A"])

    # saturation: one read repeated 70 000 times (every accepted k-mer in it reaches 65535)
    one = rand_seq(rng, 150)
    fq7 = os.path.join(work, "sat.fq")
    write_fq(fq7, [one] * 70000)
    case("L3K9_saturate", "L3K9", [fq7], ["-A"])
    # poly-A / key-0 neighbourhood and homopolymers
    fq8 = os.path.join(work, "homo.fq")
    write_fq(fq8, ["A" * 150, "C" * 150, "G" * 150, "T" * 150, "AC" * 75, "ACGT" * 40] * 50)
    case("L3K9_homopolymer", "L3K9", [fq8], ["-A"])
    case("L3K11_homopolymer", "L3K11", [fq8], ["-A"])
    # accept-everything table: collisions galore from tiny inputs
    fq9 = os.path.join(work, "dense400.fq")
    write_fq(fq9, [rand_seq(rng, 150) for _ in range(400)])
    case("L0K6_dense400_load0.4", "L0K6", [fq9], ["-A"])
    fq10 = os.path.join(work, "dense_pool.fq")
    write_fq(fq10, seqs[:4000] if False else [pool[a:a + 150] for a in (rng.randrange(0, 30000) for _ in range(6000))])
    case("L0K6_pool_counts", "L0K6", [fq10], ["-A"])
    case("L1K7_pool_counts", "L1K7", [fq10, ], ["-A"])
    case("L0K6_saturate", "L0K6", [fq7], ["-A"])
    case("L0K6z_key0", "L0K6z", [fq8], ["-A"])
    case("L0K6_ragged", "L0K6", [fq3], ["-A"], expect_abort=True)   # > hashlimit distinct keys: both must abort
    fq11 = os.path.join(work, "ragged500.fq")
    write_fq(fq11, seqs[:500])
    case("L0K6_ragged500", "L0K6", [fq11], ["-A"])
    case("L1K7_ragged", "L1K7", [fq3], ["-A"], expect_abort=True)
    case("L1K7_ragged500", "L1K7", [fq11], ["-A"])
    # gz input goes through zcat in both
    subprocess.check_call("gzip -kf %s" % fq3, shell=True)
    case("L3K9_ragged_gz", "L3K9", [fq3 + ".gz"], ["-A"])
    # 16 components
    case("L2K11_pool60k", "L2K11", [fq2], ["-A"])

    # ---- FASTQ without -A: fastq2co() + write_fqco2file() (-n minimum occurrence, -Q quality byte) ------------
    case("L3K11_syn100k_set", "L3K11", [fq], [])
    for n in ("1", "2", "3", "7", "9", "0"):           # 9 clamps to 7, 0 clamps to 1 (command_dist_wrapper.c:169-180)
        case("L3K9_pool60k_n%s" % n, "L3K9", [fq2], ["-n", n])
        case("L1K7_pool_n%s" % n, "L1K7", [fq10], ["-n", n])
    fq16 = os.path.join(work, "lowcov.fq")      # ~3x coverage with both strands: occurrence counts spread over 1..10
    lc = [pool[a:a + 150] for a in (rng.randrange(0, 60000) for _ in range(1200))]
    write_fq(fq16, [revcomp(x) if rng.random() < 0.5 else x for x in lc])
    for n in range(1, 8):
        case("L0K6_lowcov_n%d" % n, "L0K6", [fq16], ["-n", str(n)])
        case("L1K7_lowcov_n%d" % n, "L1K7", [fq16], ["-n", str(n)])
    case("L2K11_pool60k_n2", "L2K11", [fq2], ["-n", "2"])
    case("L3K9_ragged_set", "L3K9", [fq3], [])
    case("L3K9_ragged_crlf_set", "L3K9", [fq4], ["-n", "2"])
    case("L3K9_trunc_last_set", "L3K9", [fq5], [])
    case("L3K9_no_final_nl_set", "L3K9", [fq6], [])      # last record is read but never walked (iseq2comem.c:357)
    fq12 = os.path.join(work, "one_nonl.fq")
    write_fq(fq12, [rand_seq(rng, 300)], final_newline=False)   # ... except when it is the first record (:343-349)
    case("L1K7_one_record_no_nl_set", "L1K7", [fq12], [])
    case("L3K9_saturate_n7", "L3K9", [fq7], ["-n", "7"])
    case("L0K6z_key0_set", "L0K6z", [fq8], ["-n", "2"])
    case("L0K6_dense400_set", "L0K6", [fq9], [])
    case("L0K6_pool_n3", "L0K6", [fq10], ["-n", "3"])
    # fastq2co never advances its key counter (:404): no abort above hashlimit (78642 of 131071 slots here)
    fq13 = os.path.join(work, "dense700.fq")
    write_fq(fq13, [rand_seq(rng, 150) for _ in range(700)])
    case("L0K6_dense700_load0.73_set", "L0K6", [fq13], [])
    # per-base qualities: Phred+33 characters '#'..'I', threshold is the raw character code
    qseqs = [pool[a:a + 150] for a in (rng.randrange(0, 60000) for _ in range(20000))]
    quals = ["".join(rng.choice("#+5?I") if rng.random() < 0.08 else "I" for _ in range(150)) for _ in qseqs]
    fq14 = os.path.join(work, "qual.fq")
    write_fq(fq14, qseqs, quals=quals)
    for Q in ("0", "36", "54", "64", "73", "74"):
        case("L3K9_qual_Q%s" % Q, "L3K9", [fq14], ["-Q", Q])
        case("L1K7_qual_Q%s_n2" % Q, "L1K7", [fq14], ["-Q", Q, "-n", "2"])
    # long reads (fgets width 20000 there): one row each in the oracle, windows in the product
    lseqs = [rand_seq(rng, L) for L in (4094, 4095, 4096, 5000, 8191, 12000, 19997, 150, 0, 7000)]
    fq15 = os.path.join(work, "long.fq")
    write_fq(fq15, lseqs)
    case("L3K9_long_reads_set", "L3K9", [fq15], [])
    case("L1K7_long_reads_n1", "L1K7", [fq15], [])
    subprocess.check_call("gzip -kf %s" % fq14, shell=True)
    case("L3K10_qual_gz_Q54", "L3K10", [fq14 + ".gz"], ["-Q", "54"])

    # ---- FASTA cases (config 5 family): single file each so the reference's random file order is moot --
    def write_fa(path, contigs, width=70):
        with open(path, "w") as f:
            for i, c in enumerate(contigs):
                f.write(">contig_%d some description\n" % i)
                for j in range(0, len(c), width):
                    f.write(c[j:j + width] + "\n")

    g = rand_seq(rng, 400000)
    contigs = [g[:150000], g[150000:150050] + "N" * 37 + g[150050:300000], g[100000:180000], revcomp(g[300000:400000]).lower()]
    fa1 = os.path.join(work, "g1.fa")
    write_fa(fa1, contigs)
    for sh in ("L3K10", "L3K9", "L2K11"):
        case("%s_fasta" % sh, sh, [fa1], [])
        case("%s_fasta_uniq" % sh, sh, [fa1], ["-u"])
    fa3 = os.path.join(work, "g3.fasta")
    write_fa(fa3, [g[:30000], g[10000:20000], "A" * 100 + g[500:900] + "T" * 50])
    for sh in ("L0K6", "L0K6z", "L1K7"):
        case("%s_fasta_dense" % sh, sh, [fa3], [])
        case("%s_fasta_dense_uniq" % sh, sh, [fa3], ["-u"])
    case("L0K6_fasta_crowded", "L0K6", [fa1], [], expect_abort=True)
    fa2 = os.path.join(work, "g2.fna")
    write_fa(fa2, [rand_seq(rng, 30011), "ACGT" * 10, "", rand_seq(rng, 21), rand_seq(rng, 22)], width=60)
    case("L3K10_fasta_small", "L3K10", [fa2], [])
    fa4 = os.path.join(work, "g4.fa")   # the file ends inside a header line: fasta2co() gives up (iseq2comem.c:269), so does the oracle
    open(fa4, "w").write(">c0\n" + g[:5000] + "\n>unterminated header")
    case("L1K7_fasta_header_at_eof", "L1K7", [fa4], [], expect_abort=True)

    # ---- `set -u` / `set -q` on sketch directories made by the reference itself (command_set.c:241-319, 427-512) ----
    def set_case(label, shuf, inputs, dist_flags, op, reply=b"N\n"):
        nonlocal failures
        sk = os.path.join(work, label + ".sk")
        o_ref, o_ora = os.path.join(work, label + ".refpan"), os.path.join(work, label + ".orapan")
        for d in (sk, o_ref, o_ora):
            shutil.rmtree(d, ignore_errors=True)
        run([REF, "dist", "-L", shufs[shuf]] + dist_flags + ["-p", "1", "-o", sk] + inputs, work)
        r1 = subprocess.run([REF, "set", op, "-o", o_ref, sk], cwd=work, input=reply, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        r2 = subprocess.run([ORA, "set", op, "-o", o_ora, sk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        names = sorted(os.listdir(o_ref)) if os.path.isdir(o_ref) else []
        ok = r1.returncode == 0 and r2.returncode == 0 and names and names == sorted(os.listdir(o_ora))
        ok = ok and all(filecmp.cmp(os.path.join(o_ref, f), os.path.join(o_ora, f), shallow=False) for f in names)
        nids = sum(os.path.getsize(os.path.join(o_ref, f)) // 4 for f in names if f != "cofiles.stat")
        print("%s %-28s ids=%d files=%d" % ("ok  " if ok else "FAIL", label, nids, len(names)))
        failures += 0 if ok else 1

    fas = []
    for i, (a, b) in enumerate([(0, 200000), (100000, 300000), (150000, 400000), (0, 400000)]):
        pth = os.path.join(work, "strain%d.fa" % i)
        write_fa(pth, [g[a:b]])
        fas.append(pth)
    small = []
    for i, (a, b) in enumerate([(0, 20000), (10000, 30000), (15000, 40000)]):   # dense table: keep below hashlimit
        pth = os.path.join(work, "small%d.fa" % i)
        write_fa(pth, [g[a:b]])
        small.append(pth)
    for sh in ("L1K7", "L0K6z", "L3K10", "L2K11"):
        set_case("%s_set_u" % sh, sh, small if sh == "L0K6z" else fas, [], "-u")          # fas[3] covers the others
        set_case("%s_set_q" % sh, sh, small if sh == "L0K6z" else fas[:3], [], "-q")
    set_case("L1K7_set_u_reads_A", "L1K7", [fq10, fq16, fq11], ["-A"], "-u")
    set_case("L1K7_set_q_reads_n2", "L1K7", [fq10, fq16, fq11], ["-n", "2"], "-q")
    set_case("L1K7_set_u_single_N", "L1K7", [fas[0]], [], "-u")        # one sketch, prompt answered N: normal output

    # `set -i <pan>` / `set -s <pan>` (sketch_operate, :321-425): pan directories from the runs above
    def operate_case(label, sk_label, pan_label, op):
        nonlocal failures
        sk, pan = os.path.join(work, sk_label + ".sk"), os.path.join(work, pan_label + ".refpan")
        o_ref, o_ora = os.path.join(work, label + ".refop"), os.path.join(work, label + ".oraop")
        for d in (o_ref, o_ora):
            shutil.rmtree(d, ignore_errors=True)
        r1 = subprocess.run([REF, "set", op, pan, "-o", o_ref, sk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        r2 = subprocess.run([ORA, "set", op, pan, "-o", o_ora, sk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        names = sorted(os.listdir(o_ref)) if os.path.isdir(o_ref) else []
        ok = r1.returncode == 0 and r2.returncode == 0 and names and names == sorted(os.listdir(o_ora))
        ok = ok and all(filecmp.cmp(os.path.join(o_ref, f), os.path.join(o_ora, f), shallow=False) for f in names)
        nids = sum(os.path.getsize(os.path.join(o_ref, f)) // 4 for f in names if f.startswith("combco.") and ".index." not in f)
        print("%s %-28s ids=%d files=%d" % ("ok  " if ok else "FAIL", label, nids, len(names)))
        failures += 0 if ok else 1

    for sh in ("L1K7", "L0K6z", "L2K11"):
        operate_case("%s_set_i_q" % sh, "%s_set_u" % sh, "%s_set_q" % sh, "-i")   # sketches of _set_u x pan of _set_q
        operate_case("%s_set_s_q" % sh, "%s_set_u" % sh, "%s_set_q" % sh, "-s")
    operate_case("L1K7_set_s_reads_A", "L1K7_set_u_reads_A", "L1K7_set_q_reads_n2", "-s")   # koc=1 input, uniq_pan.N as pan
    # (the cofiles.stat header is compared byte for byte here: both sides copy the reference-made directory's bytes)

    # `set -g <tax.tsv>` (grouping_genomes, :831-974): both sides read the same reference-made directory; the category file
    # follows that directory's (time-shuffled) sketch order only in the sense that line i classifies sketch i
    def group_case(label, sk_label, tax_lines):
        nonlocal failures
        sk = os.path.join(work, sk_label + ".sk")
        taxf = os.path.join(work, label + ".tsv")
        open(taxf, "w").write("".join(t + "\n" for t in tax_lines))
        o_ref, o_ora = os.path.join(work, label + ".refgrp"), os.path.join(work, label + ".oragrp")
        for d in (o_ref, o_ora):
            shutil.rmtree(d, ignore_errors=True)
        r1 = subprocess.run([REF, "set", "-g", taxf, "-o", o_ref, sk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        r2 = subprocess.run([ORA, "set", "-g", taxf, "-o", o_ora, sk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        names = sorted(os.listdir(o_ref)) if os.path.isdir(o_ref) else []
        ok = r1.returncode == 0 and r2.returncode == 0 and names and names == sorted(os.listdir(o_ora))
        for f in names:
            if f == "cofiles.stat":  # names: the reference leaves the bytes behind each NUL uninitialised
                ok = ok and parse_stat(os.path.join(o_ref, f)) == parse_stat(os.path.join(o_ora, f))
            else:
                ok = ok and filecmp.cmp(os.path.join(o_ref, f), os.path.join(o_ora, f), shallow=False)
        st = parse_stat(os.path.join(o_ref, "cofiles.stat")) if names else {}
        print("%s %-28s ids=%s taxa=%s" % ("ok  " if ok else "FAIL", label, st.get("all_ctx"), st.get("names")))
        failures += 0 if ok else 1

    group_case("L1K7_set_g", "L1K7_set_u", ["562\tEscherichia coli", "1280", "562\tEscherichia coli", "0\tskip me"])
    group_case("L0K6z_set_g", "L0K6z_set_u", ["5\tfive", "5\tfive", "6"])
    group_case("L2K11_set_g", "L2K11_set_u", ["77\ta b c", "78\td", "77\ta b c", "78\td"])
    group_case("L1K7_set_g_reads_A", "L1K7_set_u_reads_A", ["9", "9", "10\tten"])
    # a taxon WITHOUT a k-mer in some component (three genomes of a few kilobases over L2K11's 16 components): the reference evaluates
    # LOG2(0) there (command_set.c:878) and writes an empty block for that component
    tiny = []
    for i, (a, b) in enumerate([(0, 3000), (5000, 9000), (20000, 22500)]):
        pth = os.path.join(work, "tiny%d.fa" % i)
        write_fa(pth, [g[a:b]])
        tiny.append(pth)
    set_case("L2K11_tiny_set_u", "L2K11", tiny, [], "-u")
    group_case("set_g_empty_component", "L2K11_tiny_set_u", ["1\tone", "2", "3\tthree"])

    # `composite -r <db> -q <qry> [-b]` (get_species_abundance, command_composite.c:446-649): marker db from the grouped
    # directory above by the reference's set -q / -i, queries = the three -A read sketches; stdout and .abv compared
    def composite_case(label, grp_label, qry_label):
        nonlocal failures
        grp, qsk = os.path.join(work, grp_label + ".refgrp"), os.path.join(work, qry_label + ".sk")
        uq, db = os.path.join(work, label + ".uq"), os.path.join(work, label + ".db")
        for d in (uq, db, os.path.join(work, label + ".abv_ref"), os.path.join(work, label + ".abv_ora")):
            shutil.rmtree(d, ignore_errors=True)
        subprocess.run([REF, "set", "-q", "-o", uq, grp], cwd=work, input=b"N\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        subprocess.run([REF, "set", "-i", uq, "-o", db, grp], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        r1 = subprocess.run([REF, "composite", "-r", db, "-q", qsk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        r2 = subprocess.run([ORA, "composite", "-r", db, "-q", qsk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        ok = r1.returncode == 0 and r2.returncode == 0 and r1.stdout == r2.stdout and len(r1.stdout) > 0
        a_ref, a_ora = os.path.join(work, label + ".abv_ref"), os.path.join(work, label + ".abv_ora")
        subprocess.run([REF, "composite", "-r", db, "-q", qsk, "-b", "-o", a_ref], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        subprocess.run([ORA, "composite", "-r", db, "-q", qsk, "-b", "-o", a_ora], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        names = sorted(os.listdir(a_ref)) if os.path.isdir(a_ref) else []
        ok = ok and names and names == sorted(os.listdir(a_ora))
        ok = ok and all(filecmp.cmp(os.path.join(a_ref, f), os.path.join(a_ora, f), shallow=False) for f in names)
        print("%s %-28s lines=%d abv=%d" % ("ok  " if ok else "FAIL", label, len(r1.stdout.splitlines()), len(names)))
        failures += 0 if ok else 1

    # reads of fq10 / fq16 were drawn from `pool`, not from the strains: make a db whose taxa ARE those read sets' sources
    pfa = []
    for i, (a, b) in enumerate([(0, 30000), (20000, 60000)]):
        pth = os.path.join(work, "poolpart%d.fa" % i)
        write_fa(pth, [pool[a:b]])
        pfa.append(pth)
    run([REF, "dist", "-L", shufs["L1K7"], "-p", "1", "-o", os.path.join(work, "L1K7_pooldb.sk")] + pfa, work)
    group_case("L1K7_pooldb", "L1K7_pooldb", ["1\tfirst", "2\tsecond"])
    composite_case("L1K7_composite", "L1K7_pooldb", "L1K7_set_u_reads_A")


    # ---- stage II (co2mco.c:12-87) and `dist -r` (command_dist.c:902-1079, :1531-1690): opt-in, 32 GiB per database ----
    if args.search:
        def search_db(label, shuf, ref_files, qry_sets):
            nonlocal failures
            names = [os.path.basename(f) for f in ref_files]
            sk, rmco, omco = label + ".sk", label + ".refmco", label + ".oramco"
            for d in (sk, rmco, omco):
                shutil.rmtree(os.path.join(work, d), ignore_errors=True)
            run([ORA, "-L", shufs[shuf], "-o", sk] + names, work)
            r1 = subprocess.run([REF, "dist", "-o", rmco, sk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            r2 = subprocess.run([ORA, "stage2", "-o", omco, sk] + ([] if args.dense else ["--no-index"]), cwd=work, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE)
            files = ["mco.0", "mcofiles.stat"] + (["mco.index.0"] if args.dense else [])
            ok = r1.returncode == 0 and r2.returncode == 0 and all(
                filecmp.cmp(os.path.join(work, rmco, f), os.path.join(work, omco, f), shallow=False) for f in files)
            print("%s %-28s gids=%d %s" % ("ok  " if ok else "FAIL", label + "_stage2", os.path.getsize(os.path.join(work, rmco, "mco.0")) // 4,
                                           "dense index compared" if args.dense else ""))
            failures += 0 if ok else 1
            if args.dense:
                os.remove(os.path.join(work, omco, "mco.index.0"))
            for qlabel, qflags, qfiles in qry_sets:
                qsk = "%s_%s.qsk" % (label, qlabel)
                shutil.rmtree(os.path.join(work, qsk), ignore_errors=True)
                run([ORA, "-L", shufs[shuf]] + qflags + ["-o", qsk] + [os.path.basename(f) for f in qfiles], work)
                for flags in ([], ["-M", "1"], ["-O", "0"], ["-O", "1", "-M", "1"], ["-N", "1"], ["-N", "3", "-M", "1"], ["-D", "0.05"],
                              ["--correction", "1"], ["--correction", "1", "-M", "1", "-N", "2", "-D", "0.2"], ["-N", "%d" % (len(names) + 1)]):
                    o1, o2, o3 = (os.path.join(work, "%s_%s.%s" % (label, qlabel, x)) for x in ("refout", "oraout", "oraout2"))
                    for d in (o1, o2, o3):
                        shutil.rmtree(d, ignore_errors=True)
                    a = subprocess.run([REF, "dist", "-r", rmco, "-o", o1, "--keepskf"] + flags + [qsk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                    b = subprocess.run([ORA, "search", "-r", rmco, "-o", o2, "--keepskf"] + flags + [qsk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                    c = subprocess.run([ORA, "search", "--refco", "-r", sk, "-o", o3, "--keepskf"] + flags + [qsk], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                    if flags[:1] == ["-N"] and int(flags[1]) > len(names):   # the reference gives up after the header (:1574); the oracle reports an error
                        ok = b.returncode != 0 and c.returncode != 0 and len(open(os.path.join(o1, "distance.out")).read().splitlines()) == 1
                        lines = -1
                    else:
                        ok = a.returncode == 0 and b.returncode == 0 and c.returncode == 0
                        for f in ("distance.out", "sharedk_ct.dat"):
                            ok = ok and filecmp.cmp(os.path.join(o1, f), os.path.join(o2, f), shallow=False) and \
                                filecmp.cmp(os.path.join(o1, f), os.path.join(o3, f), shallow=False)
                        lines = len(open(os.path.join(o1, "distance.out")).read().splitlines()) if ok else -1
                    print("%s %-28s %-40s lines=%d" % ("ok  " if ok else "FAIL", "%s_%s" % (label, qlabel), " ".join(flags), lines))
                    failures += 0 if ok else 1
            os.remove(os.path.join(work, rmco, "mco.index.0"))

        tiny = os.path.join(work, "tiny.fa")     # shorter than one k-mer: an empty sketch among the queries
        write_fa(tiny, [g[:10]])
        search_db("L1K7_search", "L1K7", small, [("fa", [], [small[1], fas[0], tiny]), ("koc", ["-A"], [fq10, fq16])])
        search_db("L3K10_search", "L3K10", fas, [("fa", [], [fas[2], tiny, small[0], fa1])])

    if args.big:
        fqb = os.path.join(work, "syn1p5m.fq")
        lib.mk_synth_fastq_write(fqb.encode(), 7, 0, 1500000, 150)
        case("L3K9_syn1p5m_load0.37", "L3K9", [fqb], ["-A"])

    print("%d failure(s); work dir %s" % (failures, work))
    if not args.keep:
        shutil.rmtree(work, ignore_errors=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
