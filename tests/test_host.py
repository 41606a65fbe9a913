"""host-side C layer (no GPU): .shuf format, derived parameters, FASTQ framing, FASTA windowing, sketch-dir writer"""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import util_inputs as ui
from conftest import SHUF_SPECS


@pytest.fixture(scope="module")
def capi():
    from metakssd_amd import capi as c
    return c


def test_shuf_roundtrip_and_permutation(capi, tmp_path):
    s = capi.Shuf.generate(9, 4, 2, 42)
    t = s.table.copy()
    assert sorted(t.tolist()) == list(range(16 ** 4))
    p = str(tmp_path / "x.shuf")
    s.write(p)
    assert os.path.getsize(p) == 16 + 4 * 16 ** 4  # command_shuffle.c:205-206
    hdr = struct.unpack("<4i", open(p, "rb").read(16))
    assert hdr[1:] == (9, 4, 2)
    r = capi.Shuf.read(p)
    assert (r.c.id, r.c.k, r.c.subk, r.c.drlevel) == (s.c.id, 9, 4, 2)
    assert np.array_equal(r.table, t)
    with pytest.raises(capi.MkError):
        capi.Shuf.read(str(tmp_path / "x.txt"))  # must end in .shuf (command_shuffle.c:217-219)
    assert np.array_equal(capi.Shuf.generate(9, 4, 2, 42).table, t)       # deterministic
    assert not np.array_equal(capi.Shuf.generate(9, 4, 2, 43).table, t)   # seeded


@pytest.mark.parametrize("name", sorted(SHUF_SPECS))
def test_params_match_oracle_and_survey_table(capi, name):
    """mk_params_init == oracle's restatement of get_hashsz + seq2co_global_var_initial; spot values from SURVEY 8a"""
    from oracle_binding import KoParams, load
    k, subk, drl, seed = SHUF_SPECS[name]
    sh = capi.Shuf.generate(k, subk, drl, seed) if subk < 6 else None
    if sh is None:  # avoid generating 64 MiB tables here: build the params from a fake header
        c = capi.ShufC(123, k, subk, drl, None, 16 ** subk)
        dummy = (C.c_int32 * 1)()
        c.table = C.cast(dummy, C.POINTER(C.c_int32))
        p = capi.ParamsC()
        assert capi.lib.mk_params_init(C.byref(c), C.byref(p)) == 0
    else:
        p = sh.params()
    ko = KoParams()
    assert load().ko_params_derive(123, k, subk, drl, C.byref(ko)) == 0
    for f in ("half_outctx_len", "TL", "crvsaddmove", "component_num", "comp_code_bits", "dim_start", "dim_end",
              "hashsize", "hashlimit", "tupmask", "domask", "undomask"):
        assert getattr(p, f) == getattr(ko, f), f
    if name == "L3K11":
        assert (p.hashsize, p.hashlimit, p.component_num, p.dim_end, p.TL) == (33554393, 20132635, 1, 4096, 22)
        assert p.tupmask == 2 ** 44 - 1 and p.domask == (2 ** 24 - 1) << 10 and p.undomask == (2 ** 10 - 1) << 34
    if name == "L2K11":
        assert (p.hashsize, p.component_num, p.comp_code_bits) == (536870909, 16, 4)
    if name == "L3K10":
        assert p.hashsize == 2097143


def test_params_reject_out_of_range_primer_index(capi):
    c = capi.ShufC(1, 5, 3, 2, None, 16 ** 3)  # 4*(5-2)-15 < 0 : the reference aborts (command_dist.c:291-303)
    dummy = (C.c_int32 * 1)()
    c.table = C.cast(dummy, C.POINTER(C.c_int32))
    p = capi.ParamsC()
    assert capi.lib.mk_params_init(C.byref(c), C.byref(p)) == capi.MK_ERR_FORMAT


def test_synth_rows_formula(capi):
    """read i, base b = "ACGT"[(mix64(mix64(seed ^ i) + b//32) >> 2*(b%32)) & 3]  (independent numpy restatement)"""
    def mix64(z):
        z = (z + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        return z ^ (z >> 31)
    rows = capi.synth_rows_host(77, 5, 3, 150, 160).reshape(3, 160)
    for r in range(3):
        want = bytearray()
        for b in range(150):
            w = mix64((mix64(77 ^ (5 + r)) + b // 32) & (2 ** 64 - 1))
            want.append(b"ACGT"[(w >> (2 * (b % 32))) & 3])
        assert bytes(rows[r, :150]) == bytes(want)
        assert rows[r, 150] == 10 and not rows[r, 151:].any()


def oracle_rows_result(oracle, rows, stride):
    rc, res = oracle.koc_from_rows(rows, stride)
    assert rc == 0
    return res


@pytest.mark.parametrize("variant", ["plain", "crlf", "trunc", "nonl"])
def test_fastq_framing_equals_reference_reader(capi, shufs, oracle_for, variant):
    """rows framed by mk_fastq_frame, walked by the oracle's per-read loop == the oracle's own 4x-fgets reader"""
    rs = np.random.RandomState(3)
    seqs = ui.ragged_reads(rs, 400)
    data = ui.fastq_bytes(seqs, crlf=variant == "crlf", final_newline=variant != "nonl", drop_last_qual=variant == "trunc")
    ora = oracle_for(shufs("L1K7"))
    rc, want = ora.koc_from_fastq(data)
    assert rc == 0
    rows, n, used, rc = capi.fastq_frame(data, 304)
    assert rc == 0 and used == len(data)
    assert n == ora.last_nreads  # the records the reference's 4x-fgets reader keeps
    assert n in (len(seqs), len(seqs) - 1)
    got = oracle_rows_result(ora, rows, 304)
    for (gi, gc_), (wi, wc) in zip(got, want):
        assert np.array_equal(gi, wi) and np.array_equal(gc_, wc)


def test_fastq_framing_streams_in_chunks(capi):
    rs = np.random.RandomState(4)
    seqs = ui.ragged_reads(rs, 300)
    data = ui.fastq_bytes(seqs)
    whole, n, _, rc = capi.fastq_frame(data, 304)
    assert rc == 0
    pieces, pos, buf = [], 0, b""
    while pos < len(data) or buf:
        take = data[pos:pos + 777]
        pos += len(take)
        buf += take
        final = pos >= len(data)
        rows, k, used, rc = capi.fastq_frame(buf, 304, final=final)
        assert rc == 0
        pieces.append(rows)
        buf = buf[used:]
        if final:
            break
    assert np.array_equal(np.concatenate(pieces), whole)


def test_fastq_framing_stride_too_small_and_overlong_lines(capi):
    data = ui.fastq_bytes([b"ACGT" * 10, b"A" * 200, b"ACGT"])
    rows, n, used, rc = capi.fastq_frame(data, 64)
    assert rc == capi.MK_ERR_ARG and n == 1          # caller re-frames from `used` with a wider stride
    rows2, n2, used2, rc2 = capi.fastq_frame(data[used:], 256)
    assert rc2 == 0 and n2 == 2
    long_line = ui.fastq_bytes([b"A" * 4095])        # fgets(…,4096) would split it: outside the contract
    assert capi.fastq_frame(long_line, 4096)[3] == capi.MK_ERR_FORMAT


def _occ_via_rows(ora, rows, stride, M):
    """ids of the keys seen >= M times, in slot order, from the oracle's counted per-read loop over framed rows:
    fastq2co inserts every key at first sight exactly like the counted loop, so the layouts coincide"""
    rc, res = ora.koc_from_rows(rows, stride)
    assert rc == 0
    return [ids[cnt >= M] for ids, cnt in res]


@pytest.mark.parametrize("variant", ["plain", "crlf", "trunc", "nonl", "qual", "one_nonl", "long"])
@pytest.mark.parametrize("M", [1, 3])
def test_fastq_frame_q_equals_fastq2co_reader(capi, shufs, oracle_for, variant, M):
    """rows framed by mk_fastq_frame_q (quality mask, record rule, long-read windows), walked by the oracle's per-read
    loop and filtered by count >= M == the oracle's restatement of fastq2co()+write_fqco2file() on the file bytes"""
    rs = np.random.RandomState(8)
    quals, Q, stride = None, 0, 304
    if variant == "qual":
        seqs = ui.pool_reads(rs, 3000, 150, p_n=0.0)
        quals, Q = ui.random_quals(rs, seqs), 54
    elif variant == "one_nonl":
        seqs = [ui.rand_seq(rs, 250)]
    elif variant == "long":
        seqs = [ui.rand_seq(rs, L) for L in (4094, 4095, 4096, 9000, 150, 19997)]
        seqs = seqs + seqs[:2] + seqs[:2]
        stride = 4096
    else:
        seqs = ui.pool_reads(rs, 3000, 150) + ui.ragged_reads(rs, 150)
    data = ui.fastq_bytes(seqs, crlf=variant == "crlf", final_newline=variant not in ("nonl", "one_nonl"),
                          drop_last_qual=variant == "trunc", quals=quals)
    sh = shufs("L1K7")
    ora = oracle_for(sh)
    rc, want = ora.co_from_fastq(data, Q=Q, M=M)
    assert rc == 0
    rows, n, nrec, used, rc = capi.fastq_frame_q(data, stride, 2 * sh.c.k, qmin=Q)
    assert rc == 0 and used == len(data)
    walked = len(seqs) - (1 if variant in ("trunc", "nonl") else 0)
    assert nrec == walked
    assert (n > nrec) == (variant == "long")
    got = _occ_via_rows(ora, rows, stride, M)
    assert sum(len(w[0]) for w in want) > 0 or (variant == "one_nonl" and M > 1)
    for g, (w, _) in zip(got, want):
        assert np.array_equal(g, w)


def test_fastq_frame_q_streams_in_chunks_and_widens(capi):
    rs = np.random.RandomState(9)
    seqs = ui.ragged_reads(rs, 200)
    quals = ui.random_quals(rs, seqs)
    data = ui.fastq_bytes(seqs, quals=quals)
    whole, n, nrec, used, rc = capi.fastq_frame_q(data, 304, 14, qmin=54)
    assert rc == 0 and nrec == len(seqs) and used == len(data)
    pieces, pos, buf, recs = [], 0, b"", 0
    while True:
        take = data[pos:pos + 555]
        pos += len(take)
        buf += take
        final = pos >= len(data)
        rows, k, r, used, rc = capi.fastq_frame_q(buf, 304, 14, qmin=54, final=final, records_before=recs)
        assert rc == 0
        pieces.append(rows)
        recs += r
        buf = buf[used:]
        if final:
            break
    assert recs == len(seqs) and np.array_equal(np.concatenate(pieces), whole)
    # a read longer than the row: MK_ERR_ARG until the stride is 4096, then it is windowed
    data = ui.fastq_bytes([b"ACGT" * 10, b"A" * 500])
    rows, n, nrec, used, rc = capi.fastq_frame_q(data, 64, 14)
    assert rc == capi.MK_ERR_ARG and nrec == 1
    assert capi.fastq_frame_q(ui.fastq_bytes([b"A" * 19999]), 4096, 14)[4] == capi.MK_ERR_FORMAT


@pytest.mark.parametrize("variant", ["plain", "ragged", "crlf", "trunc", "nonl", "long_headers"])
@pytest.mark.parametrize("occ", [False, True])
def test_threaded_framing_equals_serial(capi, variant, occ):
    """mk_fastq_frame_mt cuts the buffer at every fourth line start and frames the slices concurrently: same rows, counts and
    consumed bytes as the serial framers, for any thread count"""
    rs = np.random.RandomState(11)
    if variant == "plain":
        seqs = [ui.rand_seq(rs, 150) for _ in range(9000)]
    elif variant == "long_headers":
        seqs = [ui.rand_seq(rs, 40) for _ in range(20000)]
    else:
        seqs = ui.ragged_reads(rs, 12000)
    quals = ui.random_quals(rs, seqs) if occ else None
    data = ui.fastq_bytes(seqs, crlf=variant == "crlf", final_newline=variant != "nonl", drop_last_qual=variant == "trunc", quals=quals)
    if variant == "long_headers":  # header lines much longer than the reads: slices start in the middle of long lines
        data = data.replace(b"@r", b"@" + b"x" * 700 + b"r")
    assert len(data) > (1 << 20)
    stride = 304
    if occ:
        want = capi.fastq_frame_q(data, stride, 14, qmin=54)
        want = (want[0], want[1], want[2], want[3], want[4])
    else:
        r = capi.fastq_frame(data, stride)
        want = (r[0], r[1], r[1], r[2], r[3])
    assert want[4] == 0
    for T in (1, 2, 3, 8, 17):
        got = capi.fastq_frame_mt(data, stride, T, occ=occ, TL=14, qmin=54)
        assert got[4] == 0 and got[1:4] == want[1:4], T
        assert np.array_equal(got[0], want[0]), T
    # not the last chunk of a stream: stops at the last complete record, like the serial call
    cutoff = len(data) - 777
    if occ:
        w2 = capi.fastq_frame_q(data[:cutoff], stride, 14, qmin=54, final=False)
        w2 = (w2[0], w2[1], w2[2], w2[3], w2[4])
    else:
        r = capi.fastq_frame(data[:cutoff], stride, final=False)
        w2 = (r[0], r[1], r[1], r[2], r[3])
    g2 = capi.fastq_frame_mt(data[:cutoff], stride, 5, occ=occ, TL=14, qmin=54, final=False)
    assert g2[4] == 0 and g2[1:4] == w2[1:4] and np.array_equal(g2[0], w2[0])
    # fewer row slots than records: the first max_rows records, consumed up to there
    g3 = capi.fastq_frame_mt(data, stride, 4, occ=occ, TL=14, qmin=54, max_rows=5000)
    assert g3[4] == 0 and g3[1] == 5000 and np.array_equal(g3[0], want[0][: 5000 * stride])
    g4 = capi.fastq_frame_mt(data[g3[3]:], stride, 4, occ=occ, TL=14, qmin=54, records_before=g3[2])
    assert g4[4] == 0 and g3[1] + g4[1] == want[1] and np.array_equal(np.concatenate([g3[0], g4[0]]), want[0])


def test_threaded_framing_reports_narrow_rows_and_long_lines(capi):
    rs = np.random.RandomState(12)
    seqs = [ui.rand_seq(rs, 150) for _ in range(8000)]
    seqs[5000] = ui.rand_seq(rs, 600)
    data = ui.fastq_bytes(seqs)
    assert capi.fastq_frame_mt(data, 304, 6)[4] == capi.MK_ERR_ARG          # caller widens the rows and calls again
    r = capi.fastq_frame_mt(data, 1024, 6)
    assert r[4] == 0 and r[1] == 8000 and np.array_equal(r[0], capi.fastq_frame(data, 1024)[0])
    seqs[5000] = ui.rand_seq(rs, 4100)
    assert capi.fastq_frame_mt(ui.fastq_bytes(seqs), 4096, 6)[4] == capi.MK_ERR_FORMAT
    # occ flavour at the maximal stride windows long reads (rows != records): the threaded call falls back to one thread
    r = capi.fastq_frame_mt(ui.fastq_bytes(seqs), 4096, 6, occ=True, TL=14)
    w = capi.fastq_frame_q(ui.fastq_bytes(seqs), 4096, 14)
    assert r[4] == 0 and r[1] == w[1] == 8001 and r[2] == 8000 and np.array_equal(r[0], w[0])


@pytest.mark.parametrize("stride,chunk", [(64, None), (256, 100), (4096, 7)])
def test_fasta_windows_cover_every_kmer_once(capi, stride, chunk):
    """rows overlap by TL-1 bases: concatenating row payloads minus the overlaps gives back the cleaned stream"""
    rs = np.random.RandomState(5)
    g = ui.rand_seq(rs, 3000)
    fa = ui.fasta_bytes([g[:1000], g[1000:1010] + b"NN" + g[1012:2000], b"", g[2000:]], width=61)
    TL = 14
    rows = capi.fasta_windows(fa, TL, stride, chunk=chunk).reshape(-1, stride)
    payloads = [bytes(r[: list(r).index(10)]) for r in rows]
    rebuilt = payloads[0]
    for p in payloads[1:]:
        assert p[: TL - 1] == rebuilt[-(TL - 1):]
        rebuilt += p[TL - 1:]
    cleaned = b""
    for line in fa.split(b"\n"):
        cleaned += b">" if line.startswith(b">") else line
    assert rebuilt == cleaned
    assert all(len(p) == stride - 1 for p in payloads[:-1])


def test_fasta_input_ending_inside_a_header_is_an_error(capi):
    """the reference gives up when the stream ends inside a '>' line (iseq2comem.c:259-271); with the newline it is fine"""
    rs = np.random.RandomState(14)
    body = ui.fasta_bytes([ui.rand_seq(rs, 500)])
    with pytest.raises(capi.MkError) as ei:
        capi.fasta_windows(body + b">tail header without newline", 14, 256)
    assert ei.value.code == capi.MK_ERR_FORMAT
    with pytest.raises(capi.MkError):
        capi.fasta_windows(body + b">tail header without newline", 14, 256, chunk=37)
    rows_ok = capi.fasta_windows(body + b">tail header with newline\n", 14, 256)  # fine: the '>' stays as one reset byte
    plain = capi.fasta_windows(body, 14, 256)
    assert rows_ok.size == plain.size and (rows_ok != plain).sum() <= 2


def test_sketchdir_writer_layout(capi, tmp_path):
    """combco.N / combco.index.N / combco.N.a / cofiles.stat exactly as run_stageI lays them out"""
    sh = capi.Shuf.generate(7, 4, 1, 7)
    p = sh.params()
    out = str(tmp_path / "sk")
    h = C.c_void_p()
    assert capi.lib.mk_sketchdir_open(out.encode(), C.byref(p), 1, 2, C.byref(h)) == 0
    for name, ids, cnt in (("a/x.fq", [5, 9, 7], [1, 2, 65535]), ("b/y.fq", [], [])):
        ids_a = (C.c_uint32 * max(1, len(ids)))(*ids)
        cnt_a = (C.c_uint16 * max(1, len(cnt)))(*cnt)
        comp = capi.ComponentC(C.cast(ids_a, C.POINTER(C.c_uint32)), C.cast(cnt_a, C.POINTER(C.c_uint16)), len(ids))
        res = capi.ResultC(1, len(ids), C.pointer(comp))
        assert capi.lib.mk_sketchdir_add(h, name.encode(), C.byref(res)) == 0
    assert capi.lib.mk_sketchdir_close(h) == 0
    assert open(os.path.join(out, "combco.0"), "rb").read() == struct.pack("<3I", 5, 9, 7)
    assert open(os.path.join(out, "combco.0.a"), "rb").read() == struct.pack("<3H", 1, 2, 65535)
    assert open(os.path.join(out, "combco.index.0"), "rb").read() == struct.pack("<3Q", 0, 3, 3)
    st = open(os.path.join(out, "cofiles.stat"), "rb").read()
    assert len(st) == 32 + 2 * 4 + 2 * 256
    assert struct.unpack_from("<IB3xiiiiQ", st, 0) == (sh.c.id & 0xffffffff, 1, 14, 2, 1, 2, 3)
    assert struct.unpack_from("<2I", st, 32) == (3, 0)
    assert st[40:40 + 256].rstrip(b"\0") == b"a/x.fq" and st[296:296 + 256].rstrip(b"\0") == b"b/y.fq"


def test_randomized_fastq_text_against_both_reference_readers():
    """tools/fuzz_framing.py: random, partly malformed FASTQ-like text (blank lines, missing lines, CRLF, no final newline)
    through the product's framers vs the oracle's restatements of mt_shortreads2koc's and fastq2co's readers"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_framing.py"), "--cases", "600", "--seed", "7"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b"600 cases, 0 mismatches" in r.stdout, r.stdout.decode(errors="replace")[-800:]


def test_cli_composite_d_prints_abv_vectors(tmp_path):
    """`composite -d x.abv` (read_abv, command_composite.c:186-210): host only, runs without a GPU.  The golden .abv files were
    written by the reference's `composite -b`; the expected text is its "%d\\t%f" of the stored (index, float) pairs."""
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "metakssd_amd", "bin", "metakssd")
    abv = os.path.join(root, "tests", "golden", "expected", "composite_two_queries_L1K7", "composite_two_queries_L1K7_qry0.fq.abv")
    raw = open(abv, "rb").read()
    want = "".join("%d\t%s\n" % (i, "%f" % struct.unpack("<f", struct.pack("<f", p))[0])
                   for i, p in struct.iter_unpack("<if", raw))
    r = subprocess.run([cli, "composite", "-d", "not_a_vector.txt", abv], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == "0th argument not_a_vector.txt is not a .abv file, skipped\n" + want and len(raw) >= 16


def test_cli_set_c_combines_pan_directories(tmp_path):
    """`set -c <pan dir>...` (combin_pans, command_set.c:515-608): host only.  Two golden pan directories (written by the
    reference's set -u / -q) become the two blocks of one combined sketch directory."""
    import shutil
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "metakssd_amd", "bin", "metakssd")
    exp = os.path.join(root, "tests", "golden", "expected")
    hdr = struct.pack("<IB3xiiiiQ", 77, 0, 14, 2, 1, 3, 999)
    dirs = []
    for name, src, pref in (("u", "set_u_strains_L1K7", "pan.0"), ("q", "set_q_strains_L1K7", "uniq_pan.0")):
        d = tmp_path / name
        d.mkdir()
        shutil.copy(os.path.join(exp, src, pref), str(d / pref))
        (d / "cofiles.stat").write_bytes(hdr)
        dirs.append(str(d))
    out = str(tmp_path / "combined")
    r = subprocess.run([cli, "set", "-c", "-o", out] + dirs, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    a = open(os.path.join(exp, "set_u_strains_L1K7", "pan.0"), "rb").read()
    b = open(os.path.join(exp, "set_q_strains_L1K7", "uniq_pan.0"), "rb").read()
    assert open(os.path.join(out, "combco.0"), "rb").read() == a + b
    assert struct.unpack("<3Q", open(os.path.join(out, "combco.index.0"), "rb").read()) == (0, len(a) // 4, (len(a) + len(b)) // 4)
    st = open(os.path.join(out, "cofiles.stat"), "rb").read()
    assert len(st) == 32 + 2 * 4 + 2 * 256
    assert struct.unpack_from("<iiQ", st, 16) == (1, 2, (len(a) + len(b)) // 4)          # comp_num, infile_num, all_ctx_ct
    assert struct.unpack_from("<2I", st, 32) == (len(a) // 4, len(b) // 4)
    assert st[40:40 + 256].split(b"\0")[0].decode() == dirs[0]
    # a pan directory of another .shuf id is refused (:556-559)
    (tmp_path / "q" / "cofiles.stat").write_bytes(struct.pack("<IB3xiiiiQ", 78, 0, 14, 2, 1, 3, 999))
    r = subprocess.run([cli, "set", "-c", "-o", out] + dirs, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"not match 0th shuf_id" in r.stderr


# ---- distance.out (host code of the `dist -r` search) ---------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_dist_print_equals_oracle_on_random_matrices(seed, tmp_path):
    """mk_dist_print (product, host C) against ko_dist_print (oracle restatement pinned on the reference): the same bytes for
    random sketch sizes / shared counts, every option combination, including empty sketches, identical sketches, shared = 0"""
    import numpy as np
    import oracle_binding as ob
    from metakssd_amd import capi
    rs = np.random.RandomState(seed)
    R, Q = int(rs.randint(1, 40)), int(rs.randint(1, 12))
    ref_ct = rs.randint(0, 5000, size=R).astype(np.uint32)
    qry_ct = rs.randint(0, 5000, size=Q).astype(np.uint32)
    ref_ct[rs.randint(0, R)] = 0 if seed % 2 else ref_ct[0]
    qry_ct[rs.randint(0, Q)] = 0
    qry_ct[0] = ref_ct[0]
    ct = np.zeros((Q, R), np.uint32)
    for q in range(Q):
        for r in range(R):
            m = int(min(qry_ct[q], ref_ct[r]))
            ct[q, r] = 0 if m == 0 or rs.rand() < 0.2 else (m if rs.rand() < 0.15 else rs.randint(0, m + 1))
    rn = ["ref/genome_%d.fa" % i for i in range(R)]
    qn = ["q%d.fq.gz" % i for i in range(Q)]
    for metric in (0, 1):
        for outfields in (0, 1, 2):
            for corr in (0, 1):
                for nb in sorted({0, 1, min(3, R), R}):
                    for dth in (1.0, 0.3, 0.02):
                        kw = dict(metric=metric, outfields=outfields, correction=corr, num_neigb=nb, dthreshold=dth)
                        a, b = str(tmp_path / "a.out"), str(tmp_path / "b.out")
                        assert capi.dist_print(a, ref_ct, qry_ct, rn, qn, ct, 2 * (6 + seed % 5), 2 * (seed % 3), **kw) == 0
                        assert ob.dist_print(b, ref_ct, qry_ct, rn, qn, ct, 2 * (6 + seed % 5), 2 * (seed % 3), **kw) == 0
                        assert open(a, "rb").read() == open(b, "rb").read(), kw


def test_dist_print_rejects_what_the_reference_gives_up_on(tmp_path):
    import numpy as np
    from metakssd_amd import capi
    args = ([5, 6], [4], ["a", "b"], ["q"], np.array([[1, 2]], np.uint32), 20, 6)
    assert capi.dist_print(str(tmp_path / "x"), *args, num_neigb=3) == capi.MK_ERR_ARG       # -N above the number of references
    assert capi.dist_print(str(tmp_path / "x"), *args, metric=2) == capi.MK_ERR_ARG
    assert capi.dist_print(str(tmp_path / "x"), *args, outfields=3) == capi.MK_ERR_ARG
    assert capi.dist_print(str(tmp_path / "x"), *args, num_neigb=2) == 0


def test_cli_combines_query_directories_like_the_reference(tmp_path):
    """`dist -o <out> <sketch dir> <sketch dir>...` = combine_queries() (command_dist.c:1718-1924), host only: appended
    combco.N, continued combco.index.N, summed header, count lists and names in order.  Compared with the compiled
    reference (oracle/_ref/metakssd) byte for byte where that binary exists; the structure is checked everywhere."""
    import shutil
    import subprocess
    import util_inputs as ui
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from metakssd_amd import capi
    cli = os.path.join(root, "metakssd_amd", "bin", "metakssd")
    ora = os.path.join(root, "oracle", "kssd_oracle_cli")
    ref = os.path.join(root, "oracle", "_ref", "metakssd")
    sp = str(tmp_path / "L2K11.shuf")
    capi.Shuf.generate(9, 5, 1, 91).write(sp)  # k - drlevel = 8: one component; the 16-component layout comes with L2K11 below
    rs = np.random.RandomState(9)
    dirs = []
    for b, ng in enumerate((3, 1, 4)):
        fas = []
        for g in range(ng):
            p = str(tmp_path / ("b%d_g%d.fa" % (b, g)))
            open(p, "wb").write(ui.fasta_bytes([ui.rand_seq(rs, int(rs.randint(3000, 9000)))]))
            fas.append(p)
        d = str(tmp_path / ("batch%d" % b))
        r = subprocess.run([ora, "-L", sp, "-o", d] + fas, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        dirs.append(d)
    # a -A directory and one from another .shuf are skipped with a message, as in the reference
    sp2 = str(tmp_path / "other.shuf")
    capi.Shuf.generate(9, 5, 1, 92).write(sp2)
    other = str(tmp_path / "other")
    assert subprocess.run([ora, "-L", sp2, "-o", other, str(tmp_path / "b0_g0.fa")], stdout=subprocess.PIPE).returncode == 0
    args = [dirs[0], dirs[1], other, dirs[2]]
    out = str(tmp_path / "combined")
    r = subprocess.run([cli, "dist", "-o", out] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert b"not match 0th shuf_id" in r.stdout
    stat = open(os.path.join(out, "cofiles.stat"), "rb").read()
    infile_num, all_ctx = struct.unpack_from("<iQ", stat, 20)
    assert infile_num == 8 and len(stat) == 32 + 8 * 260
    cts = struct.unpack_from("<8I", stat, 32)
    assert sum(cts) == all_ctx
    idx = np.fromfile(os.path.join(out, "combco.index.0"), dtype=np.uint64)
    ids = np.fromfile(os.path.join(out, "combco.0"), dtype=np.uint32)
    assert len(idx) == 9 and idx[0] == 0 and idx[-1] == len(ids) and list(np.diff(idx.astype(np.int64))) == list(cts)
    parts = [np.fromfile(os.path.join(d, "combco.0"), dtype=np.uint32) for d in dirs]
    assert np.array_equal(ids, np.concatenate(parts))
    if os.path.exists(ref):
        out_ref = str(tmp_path / "combined_ref")
        r = subprocess.run([ref, "dist", "-o", out_ref] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        for f in ("combco.0", "combco.index.0"):
            assert open(os.path.join(out, f), "rb").read() == open(os.path.join(out_ref, f), "rb").read(), f
        a, b = open(os.path.join(out, "cofiles.stat"), "rb").read(), open(os.path.join(out_ref, "cofiles.stat"), "rb").read()
        assert a[:4] == b[:4] and a[8:] == b[8:]  # bytes 4..7: koc + three padding bytes the reference leaves uninitialised
    shutil.rmtree(out)


# ---- packed rows (MK_ROWS_PACKED): the host packer against a plain model of the layout in include/metakssd_hip.h ----------------
def _pack_model(seq):
    """bytes of one read -> the 64-byte packed row: dword 0 = bases | allvalid << 16, dwords 1..10 codes (first base in the top two
    bits, (byte >> 1) & 3), bytes 44.. validity (bit j of byte w: base 8w + j is one of ACGTacgt)"""
    out = np.zeros(16, dtype=np.uint32)
    vb = np.zeros(20, dtype=np.uint8)
    allv = 1
    for i, b in enumerate(seq):
        code = (b >> 1) & 3
        ok = (b & 0xDF) == b"ACTG"[code]
        if ok:
            out[1 + i // 16] |= np.uint32(code << (30 - 2 * (i % 16)))
            vb[i // 8] |= np.uint8(1 << (i % 8))
        else:
            allv = 0
    out[0] = len(seq) | (allv << 16)
    raw = out.view(np.uint8).copy()
    raw[44:64] = vb
    return raw


@pytest.mark.parametrize("no_avx2", [False, True])
def test_pack_rows_host_matches_the_layout(no_avx2):
    import subprocess
    import sys
    if no_avx2:  # the scalar form: a fresh process, the choice is made once per process
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__ + "::test_pack_rows_host_matches_the_layout[False]"],
                           env=dict(os.environ, MK_NO_AVX2="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        return
    from metakssd_amd import capi
    rs = np.random.RandomState(71)
    alphabet = np.frombuffer(b"ACGTacgtNnRYxX-*0Z\r" + bytes([0, 255, 0x61 ^ 0x20, 0xC1, 0xE7]), dtype=np.uint8)
    stride = 160
    lens = list(range(0, 40)) + [63, 64, 65, 95, 96, 97, 127, 128, 129, 143, 144, 145, 150, 151, 152] + [int(x) for x in rs.randint(0, 153, 200)]
    rows = np.zeros(len(lens) * stride, dtype=np.uint8)
    seqs = []
    for r, n in enumerate(lens):
        if r % 3 == 0:
            seq = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, n)]
        elif r % 3 == 1:
            seq = np.frombuffer(b"ACGTacgt", np.uint8)[rs.randint(0, 8, n)].copy()
            if n:
                seq[rs.randint(0, n, max(1, n // 30))] = ord("N")
        else:
            seq = alphabet[rs.randint(0, len(alphabet), n)]
        seq = np.where(seq == 10, ord("N"), seq).astype(np.uint8)
        seqs.append(bytes(seq))
        rows[r * stride: r * stride + n] = seq
        if n < stride:
            rows[r * stride + n] = 10
    packed = capi.pack_rows_host(rows, stride)
    for r, seq in enumerate(seqs):
        got = packed[r * 64:(r + 1) * 64]
        want = _pack_model(seq)
        assert np.array_equal(got, want), "read %d (%d bases): %s" % (r, len(seq), seq[:40])
    # a row of 153 bases does not fit
    long_rows = np.full(stride, ord("A"), dtype=np.uint8)
    long_rows[153] = 10
    with pytest.raises(capi.MkError):
        capi.pack_rows_host(long_rows, stride)


def _wide_model(seq):
    """bytes of up to 240 stream bytes -> the wide row (and its extension row when a byte is no base): dword 0 = bases | allvalid << 16 |
    extension follows << 17, dwords 1..15 codes; extension row: dword 0 = 1 << 18, bytes 16..45 the validity bytes"""
    out = np.zeros(32, dtype=np.uint32)
    vb = np.zeros(30, dtype=np.uint8)
    allv = 1
    for i, b in enumerate(seq):
        code = (b >> 1) & 3
        if (b & 0xDF) == b"ACTG"[code]:
            out[1 + i // 16] |= np.uint32(code << (30 - 2 * (i % 16)))
            vb[i // 8] |= np.uint8(1 << (i % 8))
        else:
            allv = 0
    out[0] = len(seq) | (allv << 16) | ((1 - allv) << 17)
    out[16] = 1 << 18
    raw = out.view(np.uint8).copy()
    raw[64 + 16:64 + 46] = vb
    return raw[:64] if allv else raw


def _fasta_stream_model(text):
    """the base stream of a FASTA text (iseq2comem.c:240-279 as mk_fasta_window / the device walk keep it): line ends dropped, a '>'
    line reduced to its '>'; -> (stream bytes, ends inside a header)"""
    out = bytearray()
    hdr = False
    for ch in text:
        if hdr:
            if ch == 10:
                hdr = False
            continue
        if ch in (10, 13):
            continue
        if ch == 62:
            hdr = True
        out.append(ch)
    return bytes(out), hdr


def _rough_fasta(rs, n_contigs, approx_len, width=70, crlf=False, rough=True):
    parts = []
    for c in range(n_contigs):
        parts.append(b">contig_%d some description > with a second angle\n" % c if rough else b">c%d\n" % c)
        n = int(rs.randint(approx_len // 2, approx_len + 1))
        seq = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, n)].copy()
        if rough and n:
            seq[rs.randint(0, n, n // 200 + 1)] = ord("N")
            lo = int(rs.randint(0, n))
            seq[lo:lo + 37] |= 0x20  # lower case
            if n > 500:
                a = int(rs.randint(0, n - 400))
                seq[a:a + int(rs.randint(1, 300))] = ord("N")
        w = width if c % 3 else int(rs.randint(1, 200))
        lines = [bytes(seq[i:i + w]) for i in range(0, n, w)]
        if rough and len(lines) > 4:
            lines.insert(len(lines) // 2, b"")          # an empty line inside a contig
            lines[1] = lines[1][:5] + b">tail of this line is header\r" + b"X"  # a '>' in the middle of a sequence line
        parts.append((b"\r\n" if crlf else b"\n").join(lines))
        parts.append(b"\r\n" if crlf else b"\n")
    return b"".join(parts)


@pytest.mark.parametrize("no_avx2", [False, True])
def test_fasta_pack_rows_is_the_walk_cut_into_rows(no_avx2):
    """mk_fasta_pack_rows == the reference's walk (modelled above, and mk_fasta_window's rows hold the same stream) cut into rows of 152
    stream bytes at a distance of 153 - TL, each packed as _pack_model says"""
    import subprocess
    import sys
    if no_avx2:
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__ + "::test_fasta_pack_rows_is_the_walk_cut_into_rows[False]"],
                           env=dict(os.environ, MK_NO_AVX2="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        return
    from metakssd_amd import capi
    rs = np.random.RandomState(17)
    texts = [b"", b"\n", b">only a header\n", b"ACGT", b"ACGTACGTACGTACGTACGTACGTACGTACGT", b">h\nACGTACGTACGTACGTACGTACGTACGTAC\n>h2\nAC",
             b"no header at all\nACGTTGCA" * 40, b">x\n" + b"ACGT" * 38 + b"\n", b">x\n" + b"ACGT" * 38, b">x\r\n" + b"ACGTN" * 61 + b"\r\n",
             _rough_fasta(rs, 3, 900), _rough_fasta(rs, 5, 3000, crlf=True), _rough_fasta(rs, 2, 70000), _rough_fasta(rs, 1, 33000, rough=False),
             _rough_fasta(rs, 40, 60, width=7), (b">a\n" + b"A" * 32765 + b"\n" + b"C" * 100 + b"\n")]
    for TL in (4, 14, 20, 22, 32):
        step = 153 - TL
        for ti, text in enumerate(texts):
            stream, hdr = _fasta_stream_model(text)
            rows, rc = capi.fasta_pack_rows(text, TL)
            assert rc == (capi.MK_ERR_FORMAT if hdr else capi.MK_OK), (TL, ti)
            if hdr:
                continue
            want_rows = (len(stream) - TL) // step + 1 if len(stream) >= TL else 0
            assert rows.size == 64 * want_rows, (TL, ti, rows.size // 64, want_rows)
            assert want_rows <= capi.lib.mk_fasta_pack_bound(len(text), TL, capi.MK_ROWS_PACKED)
            check = range(want_rows) if want_rows < 60 else sorted(set([0, 1, 2, want_rows - 3, want_rows - 2, want_rows - 1] + [int(x) for x in rs.randint(0, want_rows, 40)]))
            for r in check:
                seq = stream[r * step: r * step + 152]
                assert np.array_equal(rows[64 * r: 64 * r + 64], _pack_model(seq)), (TL, ti, r)
            # wide rows: 240 stream bytes at a distance of 241 - TL, an extension row behind every row with a byte that is no base
            wrows, rc = capi.fasta_pack_rows(text, TL, capi.MK_ROWS_WIDE)
            assert rc == capi.MK_OK
            wstep = 241 - TL
            n_wide = (len(stream) - TL) // wstep + 1 if len(stream) >= TL else 0
            assert wrows.size // 64 <= capi.lib.mk_fasta_pack_bound(len(text), TL, capi.MK_ROWS_WIDE)
            at = 0
            for r in range(n_wide):
                want = _wide_model(stream[r * wstep: r * wstep + 240]) if (n_wide < 40 or r % 7 == 0 or r >= n_wide - 2) else None
                hdr0 = int(wrows[at:at + 4].view(np.uint32)[0])
                nrow = 2 if hdr0 & 0x20000 else 1
                if want is not None:
                    assert np.array_equal(wrows[at:at + 64 * nrow], want), (TL, ti, r)
                at += 64 * nrow
            assert at == wrows.size, (TL, ti)
    # a text that ends inside a '>' line: the reference gives up
    for bad in (b">x\nACGT\n>trailing header", b">"):
        assert capi.fasta_pack_rows(bad, 20)[1] == capi.MK_ERR_FORMAT
    # the host walker's rows carry the same stream (pitch-free comparison: concatenate what each row adds)
    text = _rough_fasta(rs, 4, 5000)
    stream, _ = _fasta_stream_model(text)
    TL, stride = 20, 160
    wrows = capi.fasta_windows(text, TL, stride)
    got = bytearray()
    for r in range(wrows.size // stride):
        row = bytes(wrows[r * stride:(r + 1) * stride]).split(b"\n")[0]
        got += row if r == 0 else row[TL - 1:]
    assert bytes(got) == stream
