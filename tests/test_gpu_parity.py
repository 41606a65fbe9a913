"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle, bit-exact.

Oracle = oracle/kssd_oracle.c, pinned against the compiled reference (oracle/check_vs_ref.py, tests/golden).
Cases follow SURVEY.md 8c: collision-order (dense tables), counts, saturation, key 0, lower case, N resets,
ragged and empty rows, FASTA windows (set / uniq), 16 components, crowded abort, multi-push and shard merge.
"""
import ctypes as C
import numpy as np
import pytest

import util_inputs as ui

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from metakssd_amd import capi as c
    if c.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests must run on the MI355X box")
    return c


_engines = {}


@pytest.fixture(scope="module")
def engine_for(capi, shufs):
    def get(name):
        if name not in _engines:
            _engines[name] = capi.Engine(shufs(name), 0)
        return _engines[name]
    yield get
    for e in _engines.values():
        e.close()
    _engines.clear()


def assert_same(got, want, label=""):
    assert len(got) == len(want), label
    for c, ((gi, gc), (wi, wc)) in enumerate(zip(got, want)):
        assert len(gi) == len(wi), "%s component %d: %d ids vs oracle %d" % (label, c, len(gi), len(wi))
        assert np.array_equal(gi, wi), "%s component %d: id order/content differs" % (label, c)
        if wc is not None:
            assert gc is not None and np.array_equal(gc, wc), "%s component %d: counts differ" % (label, c)


def run_koc(capi, eng, rows, stride, pushes=1):
    eng.begin(capi.MK_MODE_KOC)
    n = rows.size // stride
    per = (n + pushes - 1) // pushes if n else 0
    done = 0
    while done < n:
        m = min(per, n - done)
        eng.push_reads(rows[done * stride:(done + m) * stride], stride, done)
        done += m
    return eng.finish()


@pytest.mark.parametrize("name,nreads", [("L3K11", 100000), ("L3K9", 100000), ("L1K7", 3000), ("L0K6", 400),
                                         ("L3K10", 50000), ("L2K11", 20000)])
def test_synthetic_uniform_reads(capi, engine_for, shufs, oracle_for, name, nreads):
    """config 2 of BASELINE.json (100 k synthetic 150 bp reads, L3K11) and its dense-table siblings"""
    rows = capi.synth_rows_host(1, 0, nreads, 150, 160)
    got = run_koc(capi, engine_for(name), rows, 160)
    rc, want = oracle_for(shufs(name)).koc_from_rows(rows, 160)
    assert rc == 0
    assert sum(len(w[0]) for w in want) > 0
    assert_same(got, want, name)


@pytest.mark.parametrize("name", ["L0K6", "L1K7", "L3K9"])
def test_pool_reads_counts_rc_N_lowercase(capi, engine_for, shufs, oracle_for, name):
    rs = np.random.RandomState(5)
    seqs = ui.pool_reads(rs, 30000, 6000)
    rows = ui.rows_from_seqs(seqs, 160)
    got = run_koc(capi, engine_for(name), rows, 160)
    rc, want = oracle_for(shufs(name)).koc_from_rows(rows, 160)
    assert rc == 0
    assert_same(got, want, name)
    if name == "L0K6":
        assert want[0][1].max() > 5  # counts really are exercised


@pytest.mark.parametrize("stride", [304, 308, 512, 4096])
@pytest.mark.parametrize("name", ["L0K6", "L1K7"])
def test_ragged_rows_and_strides(capi, engine_for, shufs, oracle_for, name, stride):
    """empty rows, rows shorter than a k-mer, rows spanning several column blocks, non-16-byte strides"""
    rs = np.random.RandomState(stride)
    seqs = ui.ragged_reads(rs, 700 if name == "L0K6" else 3000)
    rows = ui.rows_from_seqs(seqs, stride)
    got = run_koc(capi, engine_for(name), rows, stride)
    rc, want = oracle_for(shufs(name)).koc_from_rows(rows, stride)
    assert rc == 0
    assert_same(got, want, "%s stride %d" % (name, stride))


def test_long_rows_4095_bases(capi, engine_for, shufs, oracle_for):
    rs = np.random.RandomState(77)
    seqs = [ui.rand_seq(rs, 4095) for _ in range(130)] + [b"", ui.rand_seq(rs, 22)]
    rows = ui.rows_from_seqs(seqs, 4096)
    got = run_koc(capi, engine_for("L1K7"), rows, 4096)
    rc, want = oracle_for(shufs("L1K7")).koc_from_rows(rows, 4096)
    assert rc == 0
    assert_same(got, want)


def test_row_without_newline_ends_at_stride(capi, engine_for, shufs, oracle_for):
    rs = np.random.RandomState(78)
    stride = 64
    rows = ui.ACGT[rs.randint(0, 4, size=stride * 500)].copy()  # no '\n' anywhere
    got = run_koc(capi, engine_for("L0K6"), rows, stride)
    rc, want = oracle_for(shufs("L0K6")).koc_from_rows(rows, stride)
    assert rc == 0
    assert_same(got, want)


def test_saturation_65535(capi, engine_for, shufs, oracle_for):
    rs = np.random.RandomState(9)
    one = ui.rand_seq(rs, 150)
    rows = ui.rows_from_seqs([one] * 70000, 160)
    got = run_koc(capi, engine_for("L0K6"), rows, 160)
    rc, want = oracle_for(shufs("L0K6")).koc_from_rows(rows, 160)
    assert rc == 0
    assert want[0][1].max() == 65535
    assert_same(got, want)


def test_key_zero_and_homopolymers(capi, engine_for, shufs, oracle_for):
    seqs = [b"A" * 150, b"C" * 150, b"G" * 150, b"T" * 150, b"AC" * 75, b"ACGT" * 37] * 50
    rows = ui.rows_from_seqs(seqs, 160)
    got = run_koc(capi, engine_for("L0K6z"), rows, 160)
    rc, want = oracle_for(shufs("L0K6z")).koc_from_rows(rows, 160)
    assert rc == 0
    assert 0 in want[0][0]  # key 0 is representable under -A
    assert_same(got, want)


def test_multi_push_equals_single_push(capi, engine_for, shufs, oracle_for):
    rows = capi.synth_rows_host(4, 0, 500, 150, 160)
    a = run_koc(capi, engine_for("L0K6"), rows, 160, pushes=1)
    b = run_koc(capi, engine_for("L0K6"), rows, 160, pushes=7)
    assert_same(a, b)
    rc, want = oracle_for(shufs("L0K6")).koc_from_rows(rows, 160)
    assert_same(a, want)


def test_crowded_table_is_an_error_not_an_exit(capi, engine_for):
    rows = capi.synth_rows_host(5, 0, 5000, 150, 160)  # ~ 600 k distinct 12-mers >> hashlimit 78 642
    eng = engine_for("L0K6")
    eng.begin(capi.MK_MODE_KOC)
    eng.push_reads(rows, 160, 0)
    with pytest.raises(capi.CrowdedError):
        eng.finish()
    # the engine stays usable
    rows = capi.synth_rows_host(5, 0, 100, 150, 160)
    got = run_koc(capi, eng, rows, 160)
    assert len(got[0][0]) > 0


@pytest.mark.parametrize("uniq", [False, True])
@pytest.mark.parametrize("name", ["L0K6", "L0K6z", "L1K7", "L3K10", "L2K11"])
def test_fasta_windows_set_and_uniq(capi, engine_for, shufs, oracle_for, name, uniq):
    """config 5 family: fasta2co / uniq_fasta2co semantics through overlapped windows"""
    rs = np.random.RandomState(31)
    big = name in ("L3K10", "L2K11")
    g = ui.rand_seq(rs, 400000 if big else 30000)
    if big:
        contigs = [g[:150000], g[150000:150050] + b"N" * 37 + g[150050:300000], g[100000:180000],
                   ui.revcomp(g[300000:400000]).lower()]
    else:
        contigs = [g[:20000], g[5000:12000], b"A" * 100 + g[500:900] + b"T" * 50, g[20000:20021], b"", g[20021:]]
    fa = ui.fasta_bytes(contigs)
    TL = 2 * shufs(name).c.k
    for stride, chunk in ((256, None), (512, 1000), (4096, 77777)):
        rows = capi.fasta_windows(fa, TL, stride, chunk=chunk)
        eng = engine_for(name)
        eng.begin(capi.MK_MODE_UNIQ_SET if uniq else capi.MK_MODE_SET)
        eng.push_reads(rows, stride, 0)
        got = eng.finish()
        rc, want = oracle_for(shufs(name)).co_from_fasta(fa, uniq=uniq)
        assert rc == 0
        assert_same(got, want, "%s uniq=%s stride=%d" % (name, uniq, stride))


@pytest.mark.parametrize("uniq", [False, True])
@pytest.mark.parametrize("name", ["L0K6", "L0K6z", "L1K7", "L3K10", "L3K11", "L2K11"])
def test_fasta_device_stream_equals_oracle(capi, engine_for, shufs, oracle_for, name, uniq):
    """mk_sketch_push_stream: the raw FASTA bytes go to the device, which drops line ends and header lines like fasta2co()'s
    walk (iseq2comem.c:240-279) and scans overlapping rows of the base stream; whole, and in pieces that cut lines, headers
    and k-mers anywhere"""
    rs = np.random.RandomState(32)
    big = name in ("L3K10", "L3K11", "L2K11")
    g = ui.rand_seq(rs, 400000 if big else 30000)
    if big:
        contigs = [g[:150000], g[150000:150050] + b"N" * 37 + g[150050:300000], g[100000:180000],
                   ui.revcomp(g[300000:400000]).lower()]
    else:
        contigs = [g[:20000], g[5000:12000], b"A" * 100 + g[500:900] + b"T" * 50, g[20000:20021], b"", g[20021:]]
    fa = ui.fasta_bytes(contigs)
    rc, want = oracle_for(shufs(name)).co_from_fasta(fa, uniq=uniq)
    assert rc == 0
    for piece in (None, 100000, 4099, 613):
        if piece == 613 and big:
            continue
        eng = engine_for(name)
        eng.begin(capi.MK_MODE_UNIQ_SET if uniq else capi.MK_MODE_SET)
        eng.push_stream(fa, piece=piece)
        assert_same(eng.finish(), want, "%s uniq=%s piece=%s" % (name, uniq, piece))


@pytest.mark.parametrize("name", ["L0K6", "L1K7"])
def test_fasta_device_stream_rough_text(capi, engine_for, shufs, oracle_for, name):
    """CRLF, blank lines, '>' inside a sequence line, empty header, header directly behind sequence without newline at the end of
    the file, lower case, other letters, digits -- against the oracle's restatement of the reference's byte walk"""
    rs = np.random.RandomState(33)
    a, b, c = ui.rand_seq(rs, 5000), ui.rand_seq(rs, 3000), ui.rand_seq(rs, 2000)
    texts = [
        b">h1\r\n" + a[:1000] + b"\r\n" + a[1000:2000] + b"\r\n\r\n" + a[2000:] + b"\n>\n" + b + b"\n",
        b">x\n" + a[:700] + b">inline header to the end of this line\n" + a[700:1400] + b"\n" + c.lower() + b"\n",
        a + b"\n" + b + b"\n",                                  # no header at all
        b">only header\n",
        b">h\n" + a[:30] + b"R" + a[30:90] + b"-" + a[90:400] + b"5" + a[400:] + b"\n>t\n" + c,   # no newline at the end
        b"\n\n>h\n\n" + a + b"\n\n",
        b">h\n" + b"\n".join(a[i:i + 7] for i in range(0, 3000, 7)) + b"\n",          # very short lines
        b">h\n" + a[:21] + b"\n>g\n" + a[:22] + b"\n",                          # one base short of a k-mer / exactly one (k = 6,7: more)
    ]
    ora = oracle_for(shufs(name))
    for i, fa in enumerate(texts):
        rc, want = ora.co_from_fasta(fa)
        assert rc == 0
        for piece in (None, 17, 1000):
            eng = engine_for(name)
            eng.begin(capi.MK_MODE_SET)
            eng.push_stream(fa, piece=piece)
            assert_same(eng.finish(), want, "%s text %d piece=%s" % (name, i, piece))
    # a stream that ends inside a header line: the reference gives up (iseq2comem.c:259-271)
    eng = engine_for(name)
    eng.begin(capi.MK_MODE_SET)
    eng.push_stream(b">h\n" + a[:500] + b"\n>cut off")
    with pytest.raises(capi.MkError) as ei:
        eng.finish()
    assert ei.value.code == capi.MK_ERR_FORMAT


@pytest.mark.parametrize("name", ["L3K11", "L3K10", "L3K9", "L2K11"])
def test_packed_rows_equal_oracle(capi, engine_for, shufs, oracle_for, name):
    """MK_ROWS_PACKED: 64-byte rows (2 bits a base + a validity bit a base, made by mk_pack_rows_host / the framers) through
    mk_scan_packed_kernel -- the sketch of the same reads as text rows: fixed-length reads without N (every lane in step, no validity
    test), ragged reads with N / lower case / other bytes (per-lane validity bytes), reads around the k-mer length, several pushes,
    and the occurrence flavour"""
    rs = np.random.RandomState(61)
    ora = oracle_for(shufs(name))
    eng = engine_for(name)
    P = capi.MK_PACKED_PITCH | capi.MK_ROWS_PACKED
    # A: the benchmark's rows (150 bases, uniform)
    rows = capi.synth_rows_host(5, 0, 30000, 150, 160)
    rc, want = ora.koc_from_rows(rows, 160)
    assert rc == 0
    packed = capi.pack_rows_host(rows, 160)
    for pushes in (1, 3):
        eng.begin(capi.MK_MODE_KOC)
        n, per, done = 30000, (30000 + pushes - 1) // pushes, 0
        while done < n:
            m = min(per, n - done)
            eng.push_reads(packed[done * 64:(done + m) * 64], P, done)
            done += m
        assert_same(eng.finish(), want, "%s packed uniform pushes=%d" % (name, pushes))
    # B: a pool of reads (repeats: counts), reverse complements, N, lower case, odd bytes, every length 0..152
    g = ui.rand_seq(rs, 60000)
    seqs = []
    for i in range(20000):
        n = int(rs.randint(0, 153))
        a = int(rs.randint(0, len(g) - 152))
        q = bytearray(g[a:a + n])
        if i % 3 == 0:
            q = bytearray(ui.revcomp(bytes(q)))
        if n and i % 9 == 0:
            q[int(rs.randint(0, n))] = ord("N")
        if n and i % 31 == 0:
            q[int(rs.randint(0, n))] = int(rs.choice([ord("-"), ord("R"), 0x80 | ord("A"), ord("*")]))
        if i % 13 == 0:
            q = bytearray(bytes(q).lower())
        seqs.append(bytes(q))
    rows = np.zeros(len(seqs) * 160, dtype=np.uint8)
    for i, q in enumerate(seqs):
        rows[i * 160:i * 160 + len(q)] = np.frombuffer(q, np.uint8)
        rows[i * 160 + len(q)] = 10
    rc, want = ora.koc_from_rows(rows, 160)
    assert rc == 0
    packed = capi.pack_rows_host(rows, 160)
    eng.begin(capi.MK_MODE_KOC)
    eng.push_reads(packed, P, 0)
    got = eng.finish()
    assert_same(got, want, name + " packed ragged")
    # ... and the same rows as text give the same sketch (the two kernels against each other)
    eng.begin(capi.MK_MODE_KOC)
    eng.push_reads(rows, 160, 0)
    assert_same(eng.finish(), got, name + " text rows against packed rows")
    # mixed in one sketch: text rows, then packed rows (ordinals go on)
    half = len(seqs) // 2
    eng.begin(capi.MK_MODE_KOC)
    eng.push_reads(rows[:half * 160], 160, 0)
    eng.push_reads(packed[half * 64:], P, half)
    assert_same(eng.finish(), want, name + " text then packed")
    # through the library's FASTQ stream with packed buffers: the -A flavour and the occurrence flavour (fastq2co, keys seen twice)
    data = ui.fastq_bytes([q for q in seqs if b"\x0a" not in q])
    TL = 2 * shufs(name).c.k
    rc, wantq = ora.koc_from_fastq(data)
    assert rc == 0
    eng.begin(capi.MK_MODE_KOC)
    eng.push_fastq(data, nthreads=4, chunk_bytes=1 << 18, packed=True)
    assert_same(eng.finish(), wantq, name + " FASTQ stream, packed buffers")
    rc, want2 = ora.co_from_fastq(data, Q=0, M=2)
    assert rc == 0
    eng.begin_occ(2)
    eng.push_fastq(data, nthreads=4, chunk_bytes=1 << 18, occ=True, TL=TL, qmin=0, packed=True)
    assert_same(eng.finish(), want2, name + " FASTQ stream, packed buffers, occurrence 2")


@pytest.mark.parametrize("name", ["L3K11", "L3K10", "L3K9", "L2K11"])
def test_text_rows_pitch160_ragged_equal_oracle(capi, shufs, oracle_for, name):
    """text rows of pitch 160 (the benchmark's layout) through mk_scan_kernel: the benchmark's rows, ragged rows with N / lower case /
    odd bytes / rows of 159 and 160 bases (no newline), a last tile of fewer than 64 rows, several pushes -- the oracle's sketch"""
    rs = np.random.RandomState(62)
    ora = oracle_for(shufs(name))
    eng = capi.Engine(shufs(name), 0)
    try:
        rows = capi.synth_rows_host(6, 0, 30011, 150, 160)
        rc, want = ora.koc_from_rows(rows, 160)
        assert rc == 0
        for pushes in (1, 4):
            assert_same(run_koc(capi, eng, rows, 160, pushes), want, "%s pitch 160 uniform pushes=%d" % (name, pushes))
        g = ui.rand_seq(rs, 50000)
        n = 20037
        rows = np.zeros(n * 160, dtype=np.uint8)
        for i in range(n):
            L = int(rs.choice([0, 1, 21, 22, 23, 60, 100, 149, 150, 151, 158, 159, 160])) if i % 5 else int(rs.randint(0, 161))
            a = int(rs.randint(0, len(g) - 160))
            q = bytearray(g[a:a + L])
            if i % 3 == 0:
                q = bytearray(ui.revcomp(bytes(q)))
            if L and i % 7 == 0:
                q[int(rs.randint(0, L))] = ord("N")
            if L and i % 29 == 0:
                q[int(rs.randint(0, L))] = int(rs.choice([ord("-"), ord("R"), 0x80 | ord("C"), 0]))
            if i % 11 == 0:
                q = bytearray(bytes(q).lower())
            rows[i * 160:i * 160 + L] = np.frombuffer(bytes(q), np.uint8)
            if L < 160:
                rows[i * 160 + L] = 10
        rc, want = ora.koc_from_rows(rows, 160)
        assert rc == 0
        assert_same(run_koc(capi, eng, rows, 160, 1), want, name + " pitch 160 ragged")
        assert_same(run_koc(capi, eng, rows, 160, 3), want, name + " pitch 160 ragged, three pushes")
    finally:
        eng.close()


def test_packed_rows_are_refused_where_no_tuned_kernel_exists(capi, shufs):
    eng = capi.Engine(shufs("L1K7"), 0)
    try:
        eng.begin(capi.MK_MODE_KOC)
        with pytest.raises(capi.MkError):
            eng.push_reads(np.zeros(64, np.uint8), capi.MK_PACKED_PITCH | capi.MK_ROWS_PACKED, 0)
        eng.finish()
    finally:
        eng.close()


def _batch_texts(rs, big):
    """a directory's worth of small FASTA texts: genomes of several contigs, repeats and reverse complements inside a file (keys seen
    twice: -u), rough text, tiny and empty ones, one that is only a header, a file without a header line"""
    g = ui.rand_seq(rs, 300000 if big else 24000)
    L = len(g)
    texts = []
    for i in range(9):
        a, b = (i * L) // 12, (i * L) // 12 + L // (4 if i % 3 else 9)
        contigs = [g[a:b], ui.revcomp(g[a + 100:a + 100 + (b - a) // 3]), g[a:a + 57] + b"N" * (i + 1) + g[a + 57:a + 900]]
        texts.append(ui.fasta_bytes(contigs))
    texts.append(b">h1\r\n" + g[:1000] + b"\r\n" + g[1000:2000] + b"\r\n\r\n" + g[2000:5000] + b"\n>\n" + g[7000:9000].lower() + b"\n")
    texts.append(b">x\n" + g[:700] + b">inline header to the end of this line\n" + g[700:1400] + b"\n")
    texts.append(g[300:4000] + b"\n" + g[100:2000] + b"\n")            # no header at all
    texts.append(b">only header\n")
    texts.append(b"")                                                 # empty file: an empty sketch here (the command line refuses it earlier)
    texts.append(b">h\n" + g[:21] + b"\n>g\n" + g[:22] + b"\n")
    texts.append(b">h\n" + g[500:3000] + b"\n>t\n" + g[4000:5000])     # no newline at the end
    texts.append(b">h\n" + b"\n".join(g[i:i + 7] for i in range(0, 3000, 7)) + b"\n")
    return texts


@pytest.mark.parametrize("uniq", [False, True])
@pytest.mark.parametrize("name", ["L0K6", "L0K6z", "L1K7", "L3K10", "L3K11", "L2K11"])
def test_batch_of_files_equals_oracle(capi, engine_for, shufs, oracle_for, name, uniq):
    """mk_sketch_batch_begin / _end: many files in ONE launch sequence (the reference's team over files, command_dist.c:363-372) --
    every file's sketch is the oracle's fasta2co() / uniq_fasta2co() sketch of that file alone; separate buffers and the
    single-copy layout; two batches in flight"""
    rs = np.random.RandomState(52)
    texts = _batch_texts(rs, name in ("L3K10", "L3K11", "L2K11"))
    ora = oracle_for(shufs(name))
    want = []
    for t in texts:
        rc, w = ora.co_from_fasta(t, uniq=uniq) if t else (0, None)
        assert rc == 0
        want.append(w)
    mode = capi.MK_MODE_UNIQ_SET if uniq else capi.MK_MODE_SET
    eng = engine_for(name)
    ncomp = eng.params.component_num

    def check(res, texts_, want_, label):
        assert len(res) == len(texts_)
        for i, (st, alone, comps) in enumerate(res):
            assert st == 0, "%s file %d: status %d" % (label, i, st)
            if want_[i] is None:
                assert all(len(c) == 0 for c in comps) and len(comps) == ncomp
            else:
                assert_same([(c, None) for c in comps], want_[i], "%s file %d (alone=%d)" % (label, i, alone))
    for one_buffer in (False, True):
        eng.batch_begin(texts, mode, one_buffer=one_buffer)
        check(eng.batch_end(), texts, want, "%s uniq=%s one_buffer=%s" % (name, uniq, one_buffer))
    # two batches in flight, of different sizes; then the engine's ordinary path still works
    eng.batch_begin(texts[:5], mode)
    eng.batch_begin(texts[5:], mode, one_buffer=True)
    check(eng.batch_end(), texts[:5], want[:5], name + " first of two")
    check(eng.batch_end(), texts[5:], want[5:], name + " second of two")
    eng.begin(mode)
    eng.push_stream(texts[0])
    assert_same(eng.finish(), want[0], name + " alone after the batches")


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("uniq", [False, True])
@pytest.mark.parametrize("name", ["L3K10", "L3K11", "L2K11", "L3K9"])
def test_batch_of_packed_rows_equals_oracle(capi, engine_for, shufs, oracle_for, name, uniq, wide):
    """mk_sketch_batch_begin_rows: the FASTA walk done by the host (mk_fasta_pack_rows), the files' packed rows sketched in one launch
    sequence -- read in place from one registered buffer (no copy command), or copied from separate arrays; every file's sketch is
    the oracle's fasta2co() / uniq_fasta2co() sketch of its TEXT; batches of rows and of texts in flight together.  wide: rows of 240 bases
    with extension rows (MK_ROWS_WIDE)"""
    fmt = capi.MK_ROWS_WIDE if wide else capi.MK_ROWS_PACKED
    rs = np.random.RandomState(54)
    texts = _batch_texts(rs, True)
    ora = oracle_for(shufs(name))
    eng = engine_for(name)
    TL = 2 * eng.params.k
    want, rows = [], []
    for t in texts:
        rc, w = ora.co_from_fasta(t, uniq=uniq) if t else (0, None)
        assert rc == 0
        want.append(w)
        r, rc = capi.fasta_pack_rows(t, TL, fmt)
        assert rc == 0
        rows.append(r)
    mode = capi.MK_MODE_UNIQ_SET if uniq else capi.MK_MODE_SET
    ncomp = eng.params.component_num
    if wide:
        assert any(int(r[:4].view(np.uint32)[0]) & 0x20000 for r in rows if r.size), "no extension row in the inputs"

    def check(res, want_, label):
        assert len(res) == len(want_)
        for i, (st, alone, comps) in enumerate(res):
            assert st == 0, "%s file %d: status %d" % (label, i, st)
            if want_[i] is None:
                assert all(len(c) == 0 for c in comps) and len(comps) == ncomp
            else:
                assert_same([(c, None) for c in comps], want_[i], "%s file %d (alone=%d)" % (label, i, alone))
    for pinned, gap in ((True, 64), (True, 4096 + 192), (False, 0)):
        eng.batch_begin_rows(rows, mode, pinned=pinned, fmt=fmt, gap=gap)
        check(eng.batch_end(), want, "%s uniq=%s pinned=%s gap=%d" % (name, uniq, pinned, gap))
    eng.batch_begin_rows(rows[:4], mode, pinned=True, fmt=fmt)
    eng.batch_begin(texts[4:], mode, one_buffer=True)
    check(eng.batch_end(), want[:4], name + " rows, first of two")
    check(eng.batch_end(), want[4:], name + " texts, second of two")
    if wide:
        eng.begin(mode)
        try:
            with pytest.raises(capi.MkError):                                           # wide rows are for batches only
                capi._check(capi.lib.mk_sketch_push_reads(eng.h, rows[1].ctypes.data, capi.MK_PACKED_PITCH | capi.MK_ROWS_WIDE, rows[1].size // 64, 0), eng.h)
        finally:
            eng.finish()
        return
    # pushed as ordinary packed rows the same rows give the same sketch
    big = max(range(len(texts)), key=lambda i: len(texts[i]))
    eng.begin(mode)
    eng.push_reads(rows[big], capi.MK_PACKED_PITCH | capi.MK_ROWS_PACKED, 0)
    assert_same(eng.finish(), want[big], name + " the largest file's rows pushed alone")


@pytest.mark.parametrize("name", ["L3K10", "L2K11"])
def test_twelve_batches_two_in_flight_rows_and_texts(capi, shufs, oracle_for, name):
    """twelve batches of rows and texts, two in flight, every file's sketch the oracle's; a file that is sketched alone in between"""
    rs = np.random.RandomState(56)
    texts = _batch_texts(rs, True)
    ora = oracle_for(shufs(name))
    eng = capi.Engine(shufs(name), 0)
    try:
        TL = 2 * eng.params.k
        want = []
        for t in texts:
            rc, w = ora.co_from_fasta(t) if t else (0, None)
            assert rc == 0
            want.append(w)
        wide = [capi.fasta_pack_rows(t, TL, capi.MK_ROWS_WIDE)[0] for t in texts]
        ncomp = eng.params.component_num

        def check(res, idx, label):
            assert len(res) == len(idx)
            for (st, alone, comps), i in zip(res, idx):
                assert st == 0, (label, i)
                if want[i] is None:
                    assert all(len(c) == 0 for c in comps) and len(comps) == ncomp
                else:
                    assert_same([(c, None) for c in comps], want[i], "%s file %d (alone=%d)" % (label, i, alone))
        eng.batch_begin_rows([wide[0]], capi.MK_MODE_SET, pinned=True, fmt=capi.MK_ROWS_WIDE)
        check(eng.batch_end(), [0], "first")
        order = list(range(len(texts)))
        pending = []
        for b in range(12):
            rs.shuffle(order)
            idx = order[: 1 + b % len(order)]
            if b % 3 == 2:
                eng.batch_begin([texts[i] for i in idx], capi.MK_MODE_SET, one_buffer=bool(b & 1))
            else:
                eng.batch_begin_rows([wide[i] for i in idx], capi.MK_MODE_SET, pinned=bool(b % 2 == 0), fmt=capi.MK_ROWS_WIDE)
            pending.append((list(idx), "batch %d" % b))
            if len(pending) == 2:
                i0, l0 = pending.pop(0)
                check(eng.batch_end(), i0, l0)
            if b == 6:   # an ordinary sketch between batches, one batch still in flight
                big = max(range(len(texts)), key=lambda i: len(texts[i]))
                assert len(pending) == 1
                eng.begin(capi.MK_MODE_SET)
                eng.push_stream(texts[big])
                assert_same(eng.finish(), want[big], "alone between the batches")
        while pending:
            i0, l0 = pending.pop(0)
            check(eng.batch_end(), i0, l0)
        with pytest.raises(capi.MkError):
            eng.set_option(9, 2)   # (MK_OPT_BATCH_QUEUES of round 4: retired, an unknown option now)
    finally:
        eng.close()


@pytest.mark.parametrize("wide", [False, True])
def test_batch_of_packed_rows_small_tables_and_refusals(capi, shufs, oracle_for, wide):
    """512 slots per file: the flagged files are sketched alone FROM THEIR ROWS by mk_sketch_batch_end; geometries without a packed
    scan kernel and rows that are no multiple of 64 bytes are refused"""
    rs = np.random.RandomState(55)
    texts = _batch_texts(rs, True)
    ora = oracle_for(shufs("L2K11"))
    eng = capi.Engine(shufs("L2K11"), 0)
    try:
        eng.set_option(capi.MK_OPT_BATCH_TAB_BITS, 9)
        fmt = capi.MK_ROWS_WIDE if wide else capi.MK_ROWS_PACKED
        rows = [capi.fasta_pack_rows(t, 22, fmt)[0] for t in texts]
        for pinned in (True, False):
            eng.batch_begin_rows(rows, capi.MK_MODE_SET, pinned=pinned, fmt=fmt)
            res = eng.batch_end()
            n_alone = 0
            for i, (st, alone, comps) in enumerate(res):
                assert st == 0
                n_alone += alone
                if texts[i]:
                    rc, w = ora.co_from_fasta(texts[i])
                    assert rc == 0
                    assert_same([(c, None) for c in comps], w, "file %d (alone=%d, pinned=%s)" % (i, alone, pinned))
            assert n_alone >= 3, "the small tables were meant to overflow"
        with pytest.raises(capi.MkError):
            eng.batch_begin_rows([rows[0][:100]], capi.MK_MODE_SET, fmt=fmt)
        with pytest.raises(capi.MkError):
            eng.batch_begin_rows([rows[0]], capi.MK_MODE_SET, fmt=0x10000000)
    finally:
        eng.close()
    eng = capi.Engine(shufs("L1K7"), 0)
    try:
        with pytest.raises(capi.MkError):
            eng.batch_begin_rows([capi.fasta_pack_rows(texts[0], 14)[0]], capi.MK_MODE_SET)
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["L1K7", "L2K11"])
def test_batch_small_tables_fall_back_to_alone(capi, shufs, oracle_for, name):
    """MK_OPT_BATCH_TAB_BITS 9: 512 slots per file -- files with more than 256 keys (or a probe sequence that runs out) are flagged
    on the device and sketched alone by mk_sketch_batch_end; the results do not change.  A text that ends inside a header line is
    MK_ERR_FORMAT for that file only."""
    rs = np.random.RandomState(53)
    texts = _batch_texts(rs, name == "L2K11") + [b">h\n" + ui.rand_seq(rs, 500) + b"\n>cut off"]
    ora = oracle_for(shufs(name))
    eng = capi.Engine(shufs(name), 0)
    try:
        eng.set_option(capi.MK_OPT_BATCH_TAB_BITS, 9)
        eng.batch_begin(texts, capi.MK_MODE_SET)
        res = eng.batch_end()
        assert res[-1][0] == capi.MK_ERR_FORMAT
        n_alone = 0
        for i, (st, alone, comps) in enumerate(res[:-1]):
            assert st == 0
            n_alone += alone
            if texts[i]:
                rc, w = ora.co_from_fasta(texts[i])
                assert rc == 0
                assert_same([(c, None) for c in comps], w, "%s file %d (alone=%d)" % (name, i, alone))
        assert n_alone >= 3, "the small tables were meant to overflow"
        with pytest.raises(capi.MkError):
            eng.batch_end()                                  # nothing in flight
        with pytest.raises(capi.MkError):
            eng.batch_begin(texts, capi.MK_MODE_KOC)         # FASTA flavours only
    finally:
        eng.close()


@pytest.mark.parametrize("name,sparse", [("L1K7", 0), ("L1K7", 1), ("L3K11", 0), ("L2K11", -1)])
def test_finish_in_two_halves_pipelined(capi, shufs, oracle_for, name, sparse):
    """mk_sketch_finish_begin / _end: the result of sketch i is copied to the host while sketch i + 1 is already being scanned;
    every result equals the oracle's (and so the plain mk_sketch_finish's), results larger than the staging arrays included"""
    rs = np.random.RandomState(41)
    eng = capi.Engine(shufs(name), 0, sparse=sparse)
    ora = oracle_for(shufs(name))
    try:
        eng.set_option(capi.MK_OPT_RESULT_CAP, 64)  # the first results do not fit: grown inside finish_begin
        batches = []
        for i in range(4):
            n = [3000, 0, 20000, 500][i]
            seqs = ui.pool_reads(rs, 4000, n) if n else []
            rows = ui.rows_from_seqs(seqs, 160) if n else np.zeros(0, np.uint8)
            batches.append(rows)
        wants = []
        for rows in batches:
            rc, want = ora.koc_from_rows(rows, 160)
            assert rc == 0
            wants.append(want)
        pending = None
        for i, rows in enumerate(batches):
            eng.begin(capi.MK_MODE_KOC)
            eng.push_reads(rows, 160, 0)
            if pending is not None:
                assert_same(eng.finish_end(), wants[pending], "%s sketch %d" % (name, pending))
            eng.finish_begin()
            with pytest.raises(capi.MkError):  # one result outstanding: neither finish may run before it has been taken
                eng.finish_begin()
            pending = i
        assert_same(eng.finish_end(), wants[pending], "%s last sketch" % name)
        with pytest.raises(capi.MkError):
            eng.finish_end()
        # and the plain finish still works on the same engine
        eng.begin(capi.MK_MODE_KOC)
        eng.push_reads(batches[0], 160, 0)
        assert_same(eng.finish(), wants[0], "%s plain finish afterwards" % name)
    finally:
        eng.close()


@pytest.mark.parametrize("name,cus,own_queues", [("L3K11", 32, False), ("L1K7", 32, True), ("L2K11", 64, False), ("L3K11", 128, False)])
def test_split_queues_two_engines_in_turn(capi, shufs, oracle_for, name, cus, own_queues):
    """MK_OPT_SPLIT_CUS: the scan kernel on a queue of its own, what follows it on the remaining compute units; two engines take
    sketches in turn in bench.py's call order (the next scan is queued before the wait inside the last sketch's finish_begin).  Every
    sketch equals the oracle's: several pushes a sketch, host rows and device rows, an empty sketch; bad values are refused and
    going back to one queue works"""
    hip = C.CDLL("libamdhip64.so")
    rs = np.random.RandomState(43)
    engs = [capi.Engine(shufs(name), 0), capi.Engine(shufs(name), 0)]
    dev_rows = []
    ora = oracle_for(shufs(name))
    try:
        for bad in (3, 8, 24, 48, 160, -32):
            with pytest.raises(capi.MkError):
                engs[0].set_option(capi.MK_OPT_SPLIT_CUS, bad)
        with pytest.raises(capi.MkError):  # no queue to share yet
            engs[1].share_scan_queue(engs[0])
        for e in engs:
            e.set_option(capi.MK_OPT_SPLIT_CUS, cus)
        if not own_queues:  # one case keeps a scan queue per engine
            engs[1].share_scan_queue(engs[0])
            with pytest.raises(capi.MkError):  # a borrowed queue is not lent on
                engs[0].share_scan_queue(engs[1])
            with pytest.raises(capi.MkError):  # the owner keeps its queue while it is lent
                engs[0].set_option(capi.MK_OPT_SPLIT_CUS, 0)
            assert capi.lib.mk_engine_destroy(engs[0].h) == capi.MK_ERR_STATE
        sizes = [3000, 20000, 0, 500, 64000, 7, 12000]
        batches, wants = [], []
        for n in sizes:
            rows = ui.rows_from_seqs(ui.pool_reads(rs, 4000, n), 160) if n else np.zeros(0, np.uint8)
            rc, want = ora.koc_from_rows(rows, 160)
            assert rc == 0
            batches.append(rows)
            wants.append(want)
        for b in batches:
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(max(b.size, 16))) == 0
            if b.size:
                assert hip.hipMemcpy(p, C.c_void_p(b.ctypes.data), C.c_size_t(b.size), 1) == 0
            dev_rows.append(p)
        assert hip.hipDeviceSynchronize() == 0
        pend = [None, None]

        def take(j):
            if pend[j] is not None:
                assert_same(engs[j].finish_end(), wants[pend[j]], "%s split %d sketch %d" % (name, cus, pend[j]))
                pend[j] = None

        k = len(batches)
        for i in range(k + 1):
            if i < k:
                e, rows, n = engs[i & 1], batches[i], sizes[i]
                # "nothing follows" on the last sketch (its tail on the whole device) and, against the promise, on two in the middle: an
                # engine goes to its unmasked queue and back, the results stay what they are
                e.begin(capi.MK_MODE_KOC | (capi.MK_BEGIN_NOTHING_FOLLOWS if i in (1, 3, k - 1) else 0))
                if i % 3 == 0 and n:  # device-resident rows, two pushes
                    h = n // 2
                    e.push_reads_device(dev_rows[i].value, 160, h, 0)
                    e.push_reads_device(dev_rows[i].value + h * 160, 160, n - h, h)
                elif n:
                    e.push_reads(rows, 160, 0)
            if i > 0:
                j = (i - 1) & 1
                take(j)
                engs[j].finish_begin()
                pend[j] = i - 1
        take(0)
        take(1)
        # a mark that never met a scan (an empty sketch that said "nothing follows") must not move the NEXT thing's stream: a batch keeps
        # the stream it starts on (a stale mark once sent its resolve kernel to another queue than the rest of the batch)
        rs2 = np.random.RandomState(7)
        texts = _batch_texts(rs2, name == "L3K11")[:4] if name != "L2K11" else []  # (the oracle's L2K11 table makes every file a minute)
        tw = []
        for t in texts:
            rc, w = ora.co_from_fasta(t, uniq=False) if t else (0, None)
            assert rc == 0
            tw.append(w)
        for e in engs if texts else []:
            e.begin(capi.MK_MODE_KOC | capi.MK_BEGIN_NOTHING_FOLLOWS)
            assert sum(len(c[0]) for c in e.finish()) == 0
            e.batch_begin(texts, capi.MK_MODE_SET)
            for i, (st, alone, comps) in enumerate(e.batch_end()):
                assert st == 0
                if tw[i] is None:
                    assert all(len(c) == 0 for c in comps)
                else:
                    assert_same([(c, None) for c in comps], tw[i], "%s batch behind a stale mark, file %d" % (name, i))
        # back to one queue (the borrower first); the plain finish
        engs[1].set_option(capi.MK_OPT_SPLIT_CUS, 0)
        engs[0].set_option(capi.MK_OPT_SPLIT_CUS, 0)
        engs[1].set_option(capi.MK_OPT_SPLIT_CUS, cus)
        assert_same(run_koc(capi, engs[0], batches[1], 160, pushes=3), wants[1], "%s one queue again" % name)
        assert_same(run_koc(capi, engs[1], batches[4], 160), wants[4], "%s split, plain finish" % name)
    finally:
        for e in reversed(engs):
            e.close()
        for p in dev_rows:
            hip.hipFree(p)


@pytest.mark.parametrize("front", [None, 0, 5])
def test_sparse_key_list_grows_on_demand(capi, shufs, oracle_for, front):
    """engines with sparse bookkeeping start with a short distinct-key list (32 M entries instead of hashsize: 10.7 GB at L2K11);
    a finish or an export that counts more keys than it holds grows it and compacts again (MK_OPT_KEYLIST_CAP makes that
    happen on a small table); with and without a front table, whole and merged from two shards"""
    rs = np.random.RandomState(52)
    name = "L1K7"
    eng = capi.Engine(shufs(name), 0, sparse=1, front_bits=front)
    other = capi.Engine(shufs(name), 0, sparse=1, front_bits=front)
    ora = oracle_for(shufs(name))
    hip = C.CDLL("libamdhip64.so")
    bufs = []
    try:
        rows = ui.rows_from_seqs(ui.pool_reads(rs, 100000, 8000), 160)
        rc, want = ora.koc_from_rows(rows, 160)
        assert rc == 0 and len(want[0][0]) > 1000
        for e in (eng, other):
            e.set_option(capi.MK_OPT_KEYLIST_CAP, 64)
        eng.begin(capi.MK_MODE_KOC)
        eng.push_reads(rows, 160, 0)
        assert_same(eng.finish(), want, "grown inside finish")
        # the same through an export of the second half into the first (export grows the exporting engine's list)
        half = (rows.size // 160) // 2
        for e in (eng, other):
            e.set_option(capi.MK_OPT_KEYLIST_CAP, 64)
        eng.begin(capi.MK_MODE_KOC)
        other.begin(capi.MK_MODE_KOC)
        eng.push_reads(rows[: half * 160], 160, 0)
        other.push_reads(rows[half * 160:], 160, half)
        d = other.partial_count()
        assert d > 64
        for nbytes in (8 * d, 4 * d, 8 * d):
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
            bufs.append(p)
        assert other.partial_export(bufs[0].value, bufs[1].value, bufs[2].value, d) == d
        eng.partial_import(bufs[0].value, bufs[1].value, bufs[2].value, d)
        assert_same(eng.finish(), want, "merged, both lists grown")
    finally:
        for p in bufs:
            hip.hipFree(p)
        eng.close()
        other.close()


def test_shard_merge_equals_single_engine(capi, shufs, oracle_for):
    """two engines sketch disjoint contiguous read ranges with global ordinals; the second one's
    distinct-key list is imported into the first; the merged result equals the sequential sketch"""
    import ctypes as C
    shuf = shufs("L0K6")
    rs = np.random.RandomState(41)
    seqs = ui.pool_reads(rs, 20000, 5000)
    rows = ui.rows_from_seqs(seqs, 160)
    n = len(seqs)
    cut = 2300
    e0, e1 = capi.Engine(shuf, 0), capi.Engine(shuf, 0)
    try:
        for e in (e0, e1):
            e.begin(capi.MK_MODE_KOC)
        e0.push_reads(rows[:cut * 160], 160, 0)
        e1.push_reads(rows[cut * 160:], 160, cut)
        d1 = e1.partial_count()
        assert d1 > 0
        # device buffers for the exchange (what RCCL would carry): allocate through hipMalloc via ctypes
        hip = C.CDLL("libamdhip64.so")
        bufs = []
        for nbytes in (8 * d1, 4 * d1, 8 * d1):
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
            bufs.append(p)
        got_n = e1.partial_export(bufs[0].value, bufs[1].value, bufs[2].value, d1)
        assert got_n == d1
        e0.partial_import(bufs[0].value, bufs[1].value, bufs[2].value, d1)
        merged = e0.finish()
        for p in bufs:
            hip.hipFree(p)
    finally:
        e0.close()
        e1.close()
    rc, want = oracle_for(shuf).koc_from_rows(rows, 160)
    assert rc == 0
    assert_same(merged, want)
    assert n == 5000


@pytest.mark.parametrize("name,flavour", [("L0K6", "koc"), ("L1K7", "koc"), ("L3K9", "occ2"), ("L2K11", "koc")])
@pytest.mark.parametrize("G", [2, 3, 8])
def test_slice_merge_equals_oracle(capi, shufs, oracle_for, name, flavour, G):
    """SURVEY 8e's alternative exchange through the engine's C ABI: G engines sketch contiguous read ranges with global ordinals; every
    engine cuts its list into G parts by key % G (mk_partial_export_split), starts over with empty tables (mk_partial_restart) and
    folds part g of every list (mk_partial_import); the reduced slices -- disjoint key sets -- are laid behind one another in the
    first engine's own key list (mk_partial_list_reserve / _adopt / _commit) and finished from there WITHOUT touching its table.
    The result is the oracle's sequential sketch, ids and counts in the same order."""
    import ctypes as C
    if name == "L2K11" and G != 3:
        pytest.skip("21 GB engines: one value of G is enough for the 16-component dump")
    hip = C.CDLL("libamdhip64.so")
    shuf = shufs(name)
    rs = np.random.RandomState(77 + G)
    seqs = ui.pool_reads(rs, 30000, 9000) + ui.ragged_reads(rs, 200)
    stride = 304
    if flavour == "occ2":  # fastq2co's rows: the reader's record rule and quality masking applied (mk_fastq_frame_q)
        data = ui.fastq_bytes(seqs)
        rows, n, nrec, used, rc = capi.fastq_frame_q(data, stride, 2 * shuf.c.k, qmin=0)
        assert rc == 0 and used == len(data) and n > 8000
    else:
        rows = ui.rows_from_seqs(seqs, stride)
        n = len(seqs)
    cuts = [n * g // G for g in range(G + 1)]
    engines = [capi.Engine(shuf, 0) for _ in range(1 if name == "L2K11" else G)]
    allocs = []

    def dmalloc(nbytes):
        q = C.c_void_p()
        assert hip.hipMalloc(C.byref(q), C.c_size_t(max(16, nbytes))) == 0
        allocs.append(q)
        return q.value

    def begin(e):
        e.begin_occ(2) if flavour == "occ2" else e.begin(capi.MK_MODE_KOC)
    try:
        # 1. every shard's list, cut by key % G  (with one engine for all shards -- the 21 GB geometry -- shard after shard)
        exported = []  # per shard: (keys, counts, ords pointers, part sizes)
        for g in range(G):
            e = engines[g % len(engines)]
            begin(e)
            e.push_reads(rows[cuts[g] * stride:cuts[g + 1] * stride], stride, cuts[g])
            d = e.partial_count()
            bk, bc, bo = dmalloc(8 * d), dmalloc(4 * d), dmalloc(8 * d)
            got, parts = e.partial_export_split(G, bk, bc, bo, d)
            assert got == d and sum(parts) == d and len(parts) == G
            k = np.zeros(d, np.uint64)
            assert hip.hipMemcpy(C.c_void_p(k.ctypes.data), C.c_void_p(bk), C.c_size_t(8 * d), 2) == 0
            at = 0
            for gg in range(G):  # part gg holds exactly the keys with key % G == gg
                assert np.all(k[at:at + parts[gg]] % np.uint64(G) == gg), (g, gg)
                at += parts[gg]
            assert len(np.unique(k)) == d
            exported.append((bk, bc, bo, parts))
        # 2. slice g: part g of every shard folded into empty tables; the slice's reduced list
        e0 = engines[0]
        slices = []
        for g in range(G):
            e = engines[g % len(engines)]
            if len(engines) == 1:
                begin(e)
            else:
                e.partial_restart()
            for (bk, bc, bo, parts) in exported:
                off = sum(parts[:g])
                if parts[g]:
                    e.partial_import(bk + 8 * off, bc + 4 * off, bo + 8 * off, parts[g])
            r = e.partial_count()
            lk, lc, lo = e.partial_list_reserve(r)
            ck, cc, co = dmalloc(8 * r), dmalloc(4 * r), dmalloc(8 * r)
            e.sync()
            for dst, src, w in ((ck, lk, 8), (cc, lc, 4), (co, lo, 8)):
                assert hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(w * r), 3) == 0
            # a device-to-device hipMemcpy does not wait for its copy: the engine must not get its list back (the next begin of this
            # engine overwrites it -- at once under MK_POISON, which is how this was found: four suites at a time, round 6) before it is done
            assert hip.hipDeviceSynchronize() == 0
            slices.append((ck, cc, co, r))
        # 3. the first engine: all slices behind one another in its key list, finished from the list
        if len(engines) == 1:
            begin(e0)
        else:  # its own slice is in front already: the others go behind it
            assert slices[0][3] == e0.partial_count()
        total = sum(s_[3] for s_ in slices)
        at = 0
        for g, (ck, cc, co, r) in enumerate(slices):
            if not (g == 0 and len(engines) > 1):
                e0.partial_list_adopt(ck, cc, co, r, at)
            at += r
        e0.partial_list_commit(total)
        merged = e0.finish()
    finally:
        for q in allocs:
            hip.hipFree(q)
        for e in engines:
            e.close()
    ora = oracle_for(shuf)
    if flavour == "occ2":
        rc, want = ora.co_from_fastq(data, Q=0, M=2)
    else:
        rc, want = ora.koc_from_rows(rows, stride)
    assert rc == 0
    assert_same(merged, want, "%s G=%d" % (name, G))


@pytest.mark.parametrize("merge", ["gather", "slices"])
@pytest.mark.parametrize("ndev", [2, 5])
def test_multi_library_both_merges_equal_oracle(capi, shufs, oracle_for, ndev, merge):
    """libmetakssd_multi.so in this process (capi.Multi; several engines on GPU 0, device copies instead of RCCL): rows dealt
    round-robin in pieces with global ordinals, mk_multi_finish with the gather and with the key slices -- the oracle's sketch;
    twice in a row on the same object (buffers reused), the phases' times reported"""
    shuf = shufs("L1K7")
    rs = np.random.RandomState(99)
    seqs = ui.pool_reads(rs, 40000, 12000) + ui.ragged_reads(rs, 100)
    stride = 304
    rows = ui.rows_from_seqs(seqs, stride)
    n = len(seqs)
    rc, want = oracle_for(shuf).koc_from_rows(rows, stride)
    assert rc == 0
    m = capi.Multi(shuf, [0] * ndev)
    try:
        assert m.transport() == "device copies"
        m.set_merge(capi.MK_MULTI_MERGE_GATHER if merge == "gather" else capi.MK_MULTI_MERGE_SLICES)
        for rep in range(2):
            m.begin(capi.MK_MODE_KOC)
            piece = 777
            for i, a in enumerate(range(0, n, piece)):
                b = min(n, a + piece)
                m.push_reads(i % ndev, rows[a * stride:b * stride], stride, a)
            got = m.finish()
            assert_same(got, want, "multi %s x%d rep %d" % (merge, ndev, rep))
            assert m.last_merge() == merge
            t = m.last_times()
            assert t["total_ms"] > 0 and (merge == "gather") == (t["gather_ms"] == 0.0)
        # nothing pushed at all, and rows on ONE engine only (the others' lists, and every part and slice of them, are empty)
        m.begin(capi.MK_MODE_KOC)
        got = m.finish()
        assert len(got) == 1 and len(got[0][0]) == 0
        m.begin(capi.MK_MODE_KOC)
        m.push_reads(ndev - 1, rows, stride, 0)
        assert_same(m.finish(), want, "multi %s x%d, everything on the last engine" % (merge, ndev))
    finally:
        m.close()


def test_multi_library_refuses_more_engines_than_the_slices_have_parts(capi, shufs, oracle_for):
    """the merge by key slices cuts every list into one part per engine, sixteen at most (mk_partial_export_split): a seventeenth engine is
    refused when the set is MADE, with a message that says so -- not inside the first finish (the command line's --devices takes 64 names);
    sixteen engines on one GPU work, by slices"""
    shuf = shufs("L1K7")
    with pytest.raises(capi.MkError, match="at most 16"):
        capi.Multi(shuf, [0] * 17)
    rs = np.random.RandomState(5)
    seqs = ui.pool_reads(rs, 20000, 3000)
    stride = 304
    rows = ui.rows_from_seqs(seqs, stride)
    rc, want = oracle_for(shuf).koc_from_rows(rows, stride)
    assert rc == 0
    ndev = 16
    m = capi.Multi(shuf, [0] * ndev)
    try:
        m.begin(capi.MK_MODE_KOC)
        piece = 100
        for i, a in enumerate(range(0, len(seqs), piece)):
            b = min(len(seqs), a + piece)
            m.push_reads(i % ndev, rows[a * stride:b * stride], stride, a)
        assert_same(m.finish(), want, "16 engines, auto merge")
        assert m.last_merge() == "slices"
    finally:
        m.close()


def test_device_resident_push_and_device_synth(capi, engine_for, shufs, oracle_for):
    """bench path: reads generated on the device, pushed from HBM; bytes identical to the host generator"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    nreads, stride = 2000, 160
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(nreads * stride)) == 0
    capi.synth_rows_device(0, None, 99, 1000, nreads, 150, stride, p.value)
    assert hip.hipDeviceSynchronize() == 0
    back = np.zeros(nreads * stride, np.uint8)
    assert hip.hipMemcpy(C.c_void_p(back.ctypes.data), p, C.c_size_t(back.size), 2) == 0
    host = capi.synth_rows_host(99, 1000, nreads, 150, stride)
    assert np.array_equal(back, host)
    eng = engine_for("L1K7")
    eng.begin(capi.MK_MODE_KOC)
    eng.push_reads_device(p.value, stride, nreads, 0)
    got = eng.finish()
    hip.hipFree(p)
    rc, want = oracle_for(shufs("L1K7")).koc_from_rows(host, stride)
    assert rc == 0
    assert_same(got, want)


def test_candidate_buffer_overflow_path(capi, shufs, oracle_for):
    """a wave whose candidate append buffer is full resolves hits inline: force it with a zero-capacity buffer"""
    eng = capi.Engine(shufs("L0K6"), 0, cand_cap=0)
    try:
        rs = np.random.RandomState(12)
        rows = ui.rows_from_seqs(ui.pool_reads(rs, 20000, 3000), 160)
        got = run_koc(capi, eng, rows, 160)
    finally:
        eng.close()
    rc, want = oracle_for(shufs("L0K6")).koc_from_rows(rows, 160)
    assert rc == 0
    assert_same(got, want)


def test_candidate_buffer_partial_overflow(capi, shufs, oracle_for):
    """small buffer: some candidates go through the append buffer, the rest through the inline path"""
    eng = capi.Engine(shufs("L0K6"), 0, cand_cap=100)
    try:
        rows = capi.synth_rows_host(8, 0, 500, 150, 160)
        got = run_koc(capi, eng, rows, 160)
    finally:
        eng.close()
    rc, want = oracle_for(shufs("L0K6")).koc_from_rows(rows, 160)
    assert rc == 0
    assert_same(got, want)


def test_bench_two_rank_flow_merged_equals_single(capi):
    """bench.py's N>1 code path (shard, export, exchange, import, finish) with two ranks on this one GPU and the
    lists moved by gloo; the RCCL transport itself only runs on a multi-GPU node"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = ui.free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
           "--reads-per-gpu", "1000000", "--steps", "1", "--warmup", "1", "--verify"]  # (the N > 1 t_stream leg runs too)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["merged_equals_single_engine"] is True
    assert d["config"]["distinct_keys"] > 50000
    assert d["t_stream"]["reads_per_rank"] == 1000000 and d["t_stream"]["gbases_s"] > 0 and "rank0_tail_ms" in d


@pytest.mark.parametrize("world,merge", [(2, "gather"), (3, "gather"), (3, "slices"), (4, "slices")])
def test_engine_export_import_across_processes(capi, world, merge, tmp_path):
    """the engine's export / import (and the key-slice variant's split, restart, adopt, commit) with every rank a PROCESS of its own
    holding its own engine on GPU 0, lists produced by the real scan, moved by shard.py over gloo, merged on rank 0 and checked there
    against the oracle's sequential sketch (tests/dist_gpu_worker.py)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    result = str(tmp_path / "result.txt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MK_DIST_RESULT=result, MK_DIST_MERGE=merge)
    port = ui.free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    assert open(result).read().startswith("OK")


@pytest.mark.parametrize("ranks,merge", [(2, "auto"), (4, "auto"), (3, "slices")])
def test_bench_launches_its_own_ranks(capi, ranks, merge):
    """`python3 bench.py --gpus N` exactly as a driver without a launcher would start it: no torch.distributed.run around it, no
    WORLD_SIZE.  bench.py starts its N ranks itself as a child process; on this one-GPU box they share GPU 0 and exchange through
    gloo, which the line must SAY (debug transport).  Both merges (gather below four ranks, key slices from four on), the merged
    sketch equal to one engine's, and the C product's own multi-GPU path (libmetakssd_multi.so, --inproc-multi) on the same workload
    with the same sketch"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--reads-per-gpu", "600000", "--steps", "2", "--warmup", "1",
           "--no-host-legs", "--verify", "--inproc-multi", "--merge", merge]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200, env=env)
    assert r.returncode == 0, "\n".join(l for l in r.stderr.decode(errors="replace").splitlines() if "Error" in l or "error" in l or "rank0" in l)[-4000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    want_merge = merge if merge != "auto" else ("slices" if ranks >= 4 else "gather")
    assert d["n_gpus"] == ranks and d["merged_equals_single_engine"] is True and d["merge"] == want_merge
    assert "debug transport: gloo, same device" in d["config"]["parallelism"] and d["distributed"]["distinct_devices"] == 1
    assert d["distributed"]["ranks_seen"] == list(range(ranks))
    assert set(d["rank0_tail_phases_ms"]) >= ({"gather", "import", "finish"} if want_merge == "gather" else {"export_split", "all_to_all", "fold_slice", "gather_slices", "adopt", "finish"})
    im = d["inproc_multi"]
    assert im["engines"] == ranks and im["transport"] == "device copies" and im["merge"] == want_merge
    assert im["equals_process_per_gpu_sketch"] is True and im["distinct_keys"] == d["config"]["distinct_keys"] and im["gbases_s"] > 0


def test_bench_default_flow_one_gpu(capi):
    """`python3 bench.py` at N = 1 as the driver starts it (device legs only, a small workload): the headline is ALWAYS one engine on one
    queue -- fixed before anything is timed, not the faster of two flows -- and the split-queue flow (MK_OPT_SPLIT_CUS) is a side leg timed
    for the same passes and warm-up; a setting the option does not fit, or 0, leaves the headline alone"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base = [sys.executable, os.path.join(root, "bench.py"), "--reads-per-gpu", "2000000", "--steps", "7", "--warmup", "2", "--no-host-legs",
            "--no-cpu-baseline", "--no-traffic", "--verify"]
    lines = []
    for extra in (["--no-split-leg"], ["--split-cus", "24"], ["--split-cus", "0"], []):
        r = subprocess.run(base + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
        lines.append(json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]))
    d, bad, one, auto = lines
    tr = auto["queue_flows"]
    assert tr["passes_each"] == 7 and tr["warmup_each"] == 2 and tr["headline"].startswith("one_queue")
    assert abs(auto["ms_per_step"] - tr["one_queue_ms_per_step"]) < 1e-6  # whichever flow was faster in this run
    sq = auto["split_queues"]
    assert sq["compute_units_scan"] == 256 - 32 and sq["compute_units_rest"] == 32 and abs(sq["ms_per_step"] - tr["split_ms_per_step"]) < 1e-6
    assert "one_queue" not in auto
    for x in lines:
        assert x["config"]["queues"].startswith("one engine, one queue") and x["roofline"]["compute_units"] == 256
        assert x["merged_equals_single_engine"] is True and x["steps"] == 7 and x["value"] > 0
    assert "queue_flows" not in d and "split_queues" not in d
    assert "queue_flows" not in one and "split_queues" not in one
    assert "split queues not available here" in bad["split_queues_note"] and "split_queues" not in bad
    assert d["config"]["distinct_keys"] == bad["config"]["distinct_keys"] == one["config"]["distinct_keys"] == auto["config"]["distinct_keys"] > 0


def test_bench_bare_two_ranks_is_interpretable(capi):
    """`python3 bench.py --gpus 2`, nothing else (on this one-GPU box the ranks share GPU 0 over gloo, and the line says so): BASELINE config 4
    split over the ranks, and with NO flag the line carries what a scaling curve needs -- the C product's own multi-GPU library timed
    (`inproc_multi`, with its transport), the same workload on one GPU in the same run (`same_workload_one_gpu`: ms, speedup, efficiency),
    every rank seen"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, env=env)
    assert r.returncode == 0, "\n".join(l for l in r.stderr.decode(errors="replace").splitlines() if "rror" in l or "rank0" in l)[-4000:]
    d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_reads"] == 500_000_000 and d["steps"] == 40
    assert d["distributed"]["ranks_seen"] == [0, 1]
    im = d["inproc_multi"]
    assert im["transport"] in ("device copies", "rccl") and im["engines"] == 2 and im["equals_process_per_gpu_sketch"] is True and im["ms_per_step"] > 0
    one = d["same_workload_one_gpu"]
    assert one["n_gpus"] == 2 and one["ms_per_step"] > 0 and one["sketch_equals_merged"] is True
    assert abs(one["speedup"] - one["ms_per_step"] / d["ms_per_step"]) < 1e-9 and abs(one["efficiency"] - one["speedup"] / 2) < 1e-9
    assert d["config"]["distinct_keys"] == 15_692_589  # config 4's sketch (tests/test_gpu_fullsize.py; profiles/r05_fullsize_config4_vs_reference.json)


# ---- FASTQ without -A: fastq2co + write_fqco2file through MK_MODE_OCC_SET -----------------------
def run_occ(capi, eng, data, TL, M, Q=0, stride=304, pushes=1):
    rows, n, nrec, used, rc = capi.fastq_frame_q(data, stride, TL, qmin=Q)
    assert rc == 0 and used == len(data)
    eng.begin_occ(M)
    per = (n + pushes - 1) // pushes if n else 0
    done = 0
    while done < n:
        m = min(per, n - done)
        eng.push_reads(rows[done * stride:(done + m) * stride], stride, done)
        done += m
    return eng.finish()


@pytest.mark.parametrize("M", [1, 2, 4, 7, 14])
@pytest.mark.parametrize("name", ["L0K6", "L1K7", "L3K9", "L2K11"])
def test_fastq_set_min_occurrence(capi, engine_for, shufs, oracle_for, name, M):
    """~3x coverage of a pool with both strands, N and lower case: keys seen >= M times, reference slot order"""
    if name == "L2K11" and M not in (1, 4):
        pytest.skip("4.3 GB oracle table: two values of M are enough for the 16-component dump")
    rs = np.random.RandomState(51)
    seqs = ui.pool_reads(rs, 60000, 1200) + ui.ragged_reads(rs, 100)
    data = ui.fastq_bytes(seqs)
    rc, want = oracle_for(shufs(name)).co_from_fastq(data, Q=0, M=M)
    assert rc == 0
    got = run_occ(capi, engine_for(name), data, 2 * shufs(name).c.k, M, pushes=3 if M == 2 else 1)
    assert all(g[1] is None for g in got)
    if name in ("L0K6", "L1K7") and M <= 7:
        assert sum(len(w[0]) for w in want) > 0
    assert_same(got, want, "%s M=%d" % (name, M))


@pytest.mark.parametrize("Q", [0, 36, 54, 73, 74])
def test_fastq_set_quality_threshold(capi, engine_for, shufs, oracle_for, Q):
    rs = np.random.RandomState(52)
    seqs = ui.pool_reads(rs, 60000, 1500, p_n=0.0)
    data = ui.fastq_bytes(seqs, quals=ui.random_quals(rs, seqs))
    for name, M in (("L0K6", 2), ("L1K7", 1)):
        rc, want = oracle_for(shufs(name)).co_from_fastq(data, Q=Q, M=M)
        assert rc == 0
        got = run_occ(capi, engine_for(name), data, 2 * shufs(name).c.k, M, Q=Q)
        assert (sum(len(w[0]) for w in want) == 0) == (Q == 74)
        assert_same(got, want, "%s Q=%d" % (name, Q))


@pytest.mark.parametrize("variant", ["crlf", "trunc", "nonl", "one_nonl", "long", "empty"])
def test_fastq_set_reader_edge_cases(capi, engine_for, shufs, oracle_for, variant):
    """fastq2co()'s record rule (last record unterminated: read, not walked; first record always walked), CRLF,
    reads beyond the 4096-byte row (windows overlapping by TL-1), a file of one empty read"""
    rs = np.random.RandomState(53)
    stride = 304
    if variant == "one_nonl":
        seqs = [ui.rand_seq(rs, 300)]
    elif variant == "long":
        seqs = [ui.rand_seq(rs, L) for L in (4094, 4095, 4096, 5000, 8191, 12000, 19997, 150, 0, 7000)]
        stride = 4096
    elif variant == "empty":
        seqs = [b""]
    else:
        seqs = ui.ragged_reads(rs, 400)
    data = ui.fastq_bytes(seqs, crlf=variant == "crlf", final_newline=variant not in ("nonl", "one_nonl"),
                          drop_last_qual=variant == "trunc")
    for name in ("L0K6", "L1K7"):
        if name == "L0K6" and variant == "long":
            continue  # 70 k random bases of 12-mers fit, but leave the dense table to the other cases
        rc, want = oracle_for(shufs(name)).co_from_fastq(data, Q=0, M=1)
        assert rc == 0
        got = run_occ(capi, engine_for(name), data, 2 * shufs(name).c.k, 1, stride=stride)
        assert_same(got, want, "%s %s" % (name, variant))


def test_fastq_set_key_zero_is_kept(capi, engine_for, shufs, oracle_for):
    """the 4-bit count field makes the slot word of key 0 non-zero (iseq2comem.c:398-399): unlike the FASTA set
    flavours, key 0 is an ordinary key here"""
    seqs = [b"A" * 150, b"C" * 150, b"G" * 150, b"T" * 150, b"AC" * 75, b"ACGT" * 37] * 3
    data = ui.fastq_bytes(seqs)
    for M in (1, 2, 7):
        rc, want = oracle_for(shufs("L0K6z")).co_from_fastq(data, Q=0, M=M)
        assert rc == 0
        assert M > 2 or 0 in want[0][0].tolist()
        got = run_occ(capi, engine_for("L0K6z"), data, 12, M)
        assert_same(got, want, "M=%d" % M)


def test_fastq_set_does_not_abort_above_hashlimit(capi, engine_for, shufs, oracle_for):
    """fastq2co() never advances keycount (iseq2comem.c:404): 97 k distinct keys in 131 071 slots (hashlimit 78 642)
    is a result there, not an abort; only a full table is an error here"""
    rows = capi.synth_rows_host(6, 0, 700, 150, 160)
    data = b"".join(b"@r\n" + bytes(rows[i * 160:i * 160 + 150]) + b"\n+\n" + b"I" * 150 + b"\n" for i in range(700))
    ora = oracle_for(shufs("L0K6"))
    rc, want = ora.co_from_fastq(data, Q=0, M=1)
    assert rc == 0 and len(want[0][0]) > ora.P.hashlimit
    eng = engine_for("L0K6")
    got = run_occ(capi, eng, data, 12, 1, stride=160)
    assert_same(got, want)
    eng.begin(capi.MK_MODE_KOC)       # the counted flavour still aborts there (iseq2comem.c:708-709)
    eng.push_reads(rows, 160, 0)
    with pytest.raises(capi.CrowdedError):
        eng.finish()
    big = capi.synth_rows_host(5, 0, 5000, 150, 160)  # ~600 k distinct keys: more than the table has slots
    eng.begin_occ(1)
    eng.push_reads(big, 160, 0)
    with pytest.raises(capi.CrowdedError):
        eng.finish()
    with pytest.raises(capi.MkError):
        eng.begin_occ(15)             # iseq2comem.c:325
    with pytest.raises(capi.MkError):
        eng.begin_occ(0)


def test_fastq_set_shard_merge(capi, shufs, oracle_for):
    """the multi-GPU merge carries counts, so the occurrence filter applies to the merged counts"""
    import ctypes as C
    shuf = shufs("L1K7")
    rs = np.random.RandomState(54)
    seqs = ui.pool_reads(rs, 60000, 1500)
    data = ui.fastq_bytes(seqs)
    rows, n, nrec, used, rc = capi.fastq_frame_q(data, 160, 14)
    assert rc == 0 and n == 1500
    cut = 700
    e0, e1 = capi.Engine(shuf, 0), capi.Engine(shuf, 0)
    try:
        e0.begin_occ(3)
        e1.begin_occ(3)
        e0.push_reads(rows[:cut * 160], 160, 0)
        e1.push_reads(rows[cut * 160:], 160, cut)
        d1 = e1.partial_count()
        hip = C.CDLL("libamdhip64.so")
        bufs = []
        for nbytes in (8 * d1, 4 * d1, 8 * d1):
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
            bufs.append(p)
        assert e1.partial_export(bufs[0].value, bufs[1].value, bufs[2].value, d1) == d1
        e0.partial_import(bufs[0].value, bufs[1].value, bufs[2].value, d1)
        merged = e0.finish()
        for p in bufs:
            hip.hipFree(p)
    finally:
        e0.close()
        e1.close()
    rc, want = oracle_for(shuf).co_from_fastq(data, Q=0, M=3)
    assert rc == 0 and len(want[0][0]) > 0
    assert_same(merged, want)


@pytest.mark.parametrize("world", [1, 2])
def test_genome_directory_sharded_by_file_equals_cli(capi, shufs, tmp_path, world):
    """config 5's multi-GPU shape: files sharded across ranks (here 2 ranks on the one GPU, lists through gloo),
    per-file sketches gathered in file order: the sketch directory is byte-identical to the single-process CLI's"""
    import filecmp
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = np.random.RandomState(71)
    gd = tmp_path / "genomes"
    gd.mkdir()
    for i, n in enumerate((30000, 9000, 52000, 700, 21000)):
        g = ui.rand_seq(rs, n)
        (gd / ("g%d.%s" % (i, ("fa", "fna", "fasta")[i % 3]))).write_bytes(ui.fasta_bytes([g[: n // 3], g[n // 3:]], width=60 + i))
    sp = str(tmp_path / "L1K7.shuf")
    shufs("L1K7").write(sp)
    out_cli, out_multi = str(tmp_path / "cli"), str(tmp_path / "multi")
    r = subprocess.run([os.path.join(root, "metakssd_amd", "bin", "metakssd"), "dist", "-L", sp, "-o", out_cli, str(gd)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    tool = os.path.join(root, "tools", "dist_genomes_multi.py")
    if world == 1:
        cmd = [sys.executable, tool, "-L", sp, "-o", out_multi, str(gd)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(32500 + os.getpid() % 2000), tool, "-L", sp, "-o", out_multi, "--backend", "gloo", "--same-device", str(gd)]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    names = sorted(os.listdir(out_cli))
    assert names == sorted(os.listdir(out_multi)) and "cofiles.stat" in names
    for f in names:
        assert filecmp.cmp(os.path.join(out_cli, f), os.path.join(out_multi, f), shallow=False), f


@pytest.mark.parametrize("name", ["L0K6", "L1K7", "L0K6z"])
def test_sparse_bookkeeping_over_a_sequence_of_sketches(capi, shufs, oracle_for, name):
    """tables of 2^26+ slots track which blocks were touched and clear / compact / dump only those (mk_table::dirty);
    MK_OPT_SPARSE=1 forces that on a small dense table, where one engine then runs sketches of very different sizes and
    flavours back to back -- every one must equal the oracle, i.e. nothing of an earlier sketch may survive a clear"""
    eng = capi.Engine(shufs(name), 0, sparse=1)
    ora = oracle_for(shufs(name))
    rs = np.random.RandomState(91)
    try:
        for step, nreads in enumerate([300, 3, 0, 450, 1, 200]):
            seqs = ui.pool_reads(rs, 30000, nreads) if nreads else []
            rows = ui.rows_from_seqs(seqs, 160) if nreads else np.zeros(0, np.uint8)
            got = run_koc(capi, eng, rows, 160, pushes=1 + step % 3)
            rc, want = ora.koc_from_rows(rows, 160) if nreads else (0, [(np.zeros(0, np.uint32), np.zeros(0, np.uint16))])
            assert rc == 0
            assert_same(got, want, "%s step %d" % (name, step))
            if step == 1:  # a crowded sketch in between: error, then the engine must still come back clean
                big = capi.synth_rows_host(5, 0, 30000 if name == "L1K7" else 5000, 150, 160)  # L1K7 accepts 1 k-mer in 16
                eng.begin(capi.MK_MODE_KOC)
                eng.push_reads(big, 160, 0)
                with pytest.raises(capi.CrowdedError):
                    eng.finish()
            if step == 3:  # a FASTA set sketch on the same engine
                fa = ui.fasta_bytes([ui.rand_seq(rs, 9000), ui.rand_seq(rs, 4000)])
                eng.begin(capi.MK_MODE_SET)
                eng.push_reads(capi.fasta_windows(fa, 2 * shufs(name).c.k, 256), 256, 0)
                got = eng.finish()
                rc, want = ora.co_from_fasta(fa)
                assert rc == 0
                assert_same(got, want, "%s fasta" % name)
    finally:
        eng.close()


@pytest.mark.parametrize("bits", [3, 7, 12])
@pytest.mark.parametrize("name", ["L0K6", "L1K7", "L3K9"])
def test_front_table_over_a_sequence_of_sketches(capi, shufs, oracle_for, name, bits):
    """MK_OPT_FRONT_BITS: new keys go to a small table in front of the hashsize-slot one while it is open; a key lives in the
    front table iff it is found within 16 probes there; the big table is cleared and compacted only when somebody used it.
    8 slots: nearly everything spills; 128: closes after the first launch of the larger sketches; 4096: most sketches never
    touch the big table, so its clear is skipped -- and must not be skipped after one that did.  Every sketch equals the
    oracle, across flavours, crowded and abandoned sketches."""
    eng = capi.Engine(shufs(name), 0, sparse=0, front_bits=bits)
    ora = oracle_for(shufs(name))
    rs = np.random.RandomState(93 + bits)
    try:
        for step, nreads in enumerate([300, 3, 0, 2500, 1, 200, 40, 900, 5]):
            seqs = ui.pool_reads(rs, 30000, nreads) if nreads else []
            rows = ui.rows_from_seqs(seqs, 160) if nreads else np.zeros(0, np.uint8)
            got = run_koc(capi, eng, rows, 160, pushes=1 + step % 4)
            rc, want = ora.koc_from_rows(rows, 160) if nreads else (0, [(np.zeros(0, np.uint32), np.zeros(0, np.uint16))])
            assert rc == 0
            assert_same(got, want, "%s step %d" % (name, step))
            if step == 1 and name != "L3K9":  # a crowded sketch in between
                big = capi.synth_rows_host(5, 0, 30000 if name == "L1K7" else 5000, 150, 160)
                eng.begin(capi.MK_MODE_KOC)
                eng.push_reads(big, 160, 0)
                with pytest.raises(capi.CrowdedError):
                    eng.finish()
            if step == 3:  # begun, fed (big table used), abandoned: the next begin cannot know and must clear
                eng.begin(capi.MK_MODE_KOC)
                eng.push_reads(ui.rows_from_seqs(ui.pool_reads(rs, 30000, 1500), 160), 160, 0)
            if step == 5:  # a FASTA set sketch on the same engine
                fa = ui.fasta_bytes([ui.rand_seq(rs, 9000), ui.rand_seq(rs, 4000)])
                eng.begin(capi.MK_MODE_SET)
                eng.push_reads(capi.fasta_windows(fa, 2 * shufs(name).c.k, 256), 256, 0)
                got = eng.finish()
                rc, want = ora.co_from_fasta(fa)
                assert rc == 0
                assert_same(got, want, "%s fasta" % name)
    finally:
        eng.close()


def test_front_table_shard_merge(capi, shufs, oracle_for):
    """export / import with front tables of different sizes on the two sides: the importing engine's table is part front,
    part big, the import closes the front table between its launches"""
    name = "L1K7"
    rs = np.random.RandomState(97)
    seqs = ui.pool_reads(rs, 30000, 3000)
    rows = ui.rows_from_seqs(seqs, 160)
    rc, want = oracle_for(shufs(name)).koc_from_rows(rows, 160)
    assert rc == 0
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    a = capi.Engine(shufs(name), 0, sparse=0, front_bits=6)
    b = capi.Engine(shufs(name), 0, sparse=0, front_bits=9)
    try:
        half = 1500
        a.begin(capi.MK_MODE_KOC); a.push_reads(rows[:half * 160], 160, 0)
        b.begin(capi.MK_MODE_KOC); b.push_reads(rows[half * 160:], 160, half)
        n = b.partial_count()
        bufs = [C.c_void_p() for _ in range(3)]
        for p_, sz in zip(bufs, (8, 4, 8)):
            assert hip.hipMalloc(C.byref(p_), C.c_size_t(max(1, n) * sz)) == 0
        assert b.partial_export(bufs[0].value, bufs[1].value, bufs[2].value, n) == n
        a.partial_import(bufs[0].value, bufs[1].value, bufs[2].value, n)
        a.partial_import(bufs[0].value, bufs[1].value, bufs[2].value, 0)
        got = a.finish()
        for p_ in bufs:
            hip.hipFree(p_)
        # counts of the second half were imported once
        assert_same(got, want, "front-table merge")
    finally:
        a.close(); b.close()


# ---- round 2: asynchronous pushes, the whole-file FASTQ stream bound to an engine, result-array growth, options ----------
def test_async_pushes_with_tickets(capi, shufs, oracle_for):
    """mk_sketch_push_reads_async keeps several host buffers in flight; the sketch equals the oracle's and every ticket
    can be waited for in any order (also long after its ring slot has been reused)"""
    eng = capi.Engine(shufs("L1K7"), 0)
    try:
        rs = np.random.RandomState(44)
        seqs = ui.pool_reads(rs, 40000, 4000)
        rows = ui.rows_from_seqs(seqs, 160)
        n = len(seqs)
        eng.begin(capi.MK_MODE_KOC)
        tickets, parts, per = [], [], 97
        for lo in range(0, n, per):
            part = np.ascontiguousarray(rows[lo * 160:min(n, lo + per) * 160])
            parts.append(part)  # stays alive until its ticket has been waited for
            tickets.append(eng.push_reads_async(part, 160, lo))
        assert len(tickets) > 32  # more than the engine's ticket ring
        for t in reversed(tickets):
            eng.push_wait(t)
        with pytest.raises(capi.MkError):
            eng.push_wait(tickets[-1] + 1)
        got = eng.finish()
    finally:
        eng.close()
    rc, want = oracle_for(shufs("L1K7")).koc_from_rows(rows, 160)
    assert rc == 0
    assert_same(got, want)


@pytest.mark.parametrize("occ", [False, True])
def test_engine_fastq_stream_equals_oracle(capi, shufs, oracle_for, occ):
    """mk_sketch_push_fastq: host threads frame the FASTQ text into pinned buffers, pushes go out asynchronously in file
    order; small chunks so that many buffers, strides and the serial fallback all occur"""
    rs = np.random.RandomState(8)
    seqs = ui.ragged_reads(rs, 6000) + ui.pool_reads(rs, 30000, 3000)
    quals = ui.random_quals(rs, seqs)
    data = ui.fastq_bytes(seqs, quals=quals)
    name = "L1K7"
    eng = capi.Engine(shufs(name), 0)
    ora = oracle_for(shufs(name))
    try:
        for T, chunk in ((1, 1 << 20), (6, 4096), (8, 1 << 16)):
            if occ:
                eng.begin_occ(2)
                st = eng.push_fastq(data, nthreads=T, chunk_bytes=chunk, occ=True, TL=14, qmin=54)
                got = eng.finish()
                rc, want = ora.co_from_fastq(data, Q=54, M=2)
            else:
                eng.begin(capi.MK_MODE_KOC)
                st = eng.push_fastq(data, nthreads=T, chunk_bytes=chunk)
                got = eng.finish()
                rc, want = ora.koc_from_fastq(data)
            assert rc == 0 and st.records == len(seqs)
            assert_same(got, want, "T=%d chunk=%d" % (T, chunk))
    finally:
        eng.close()


def test_result_arrays_grow_and_options_are_refused_inside_a_sketch(capi, shufs, oracle_for):
    """the dump writes into pinned host arrays; a sketch larger than they are is written a second time after they have
    grown (MK_OPT_RESULT_CAP makes them tiny here); options cannot change between begin and finish"""
    eng = capi.Engine(shufs("L0K6"), 0)
    try:
        eng.set_option(capi.MK_OPT_RESULT_CAP, 7)
        rows = capi.synth_rows_host(21, 0, 300, 150, 160)
        got = run_koc(capi, eng, rows, 160)
        rc, want = oracle_for(shufs("L0K6")).koc_from_rows(rows, 160)
        assert rc == 0 and len(want[0][0]) > 1000
        assert_same(got, want)
        got2 = run_koc(capi, eng, rows[: 160 * 5], 160)  # smaller result in the grown arrays
        rc, want2 = oracle_for(shufs("L0K6")).koc_from_rows(rows[: 160 * 5], 160)
        assert_same(got2, want2)
        eng.begin(capi.MK_MODE_KOC)
        with pytest.raises(capi.MkError):
            eng.set_option(capi.MK_OPT_SPARSE, 1)
        eng.finish()
        with pytest.raises(capi.MkError):
            eng.set_option(99, 1)
        for sparse in (1, 0, -1):  # switching the bookkeeping between sketches keeps results exact
            eng.set_option(capi.MK_OPT_SPARSE, sparse)
            assert_same(run_koc(capi, eng, rows, 160), want, "sparse=%d" % sparse)
    finally:
        eng.close()


def test_two_engines_on_two_host_threads(capi, shufs, oracle_for):
    """one engine per host thread on the same device (the multi-GPU command line drives one engine per GPU this way): the
    scan kernels' dynamic-LDS limits are kept per engine, nothing is shared between the threads"""
    import threading
    names = ["L3K11", "L3K10"]
    rows = capi.synth_rows_host(31, 0, 60000, 150, 160)
    out = {}

    def work(name):
        eng = capi.Engine(shufs(name), 0)
        try:
            for _ in range(3):
                out[name] = run_koc(capi, eng, rows, 160, pushes=3)
        finally:
            eng.close()

    th = [threading.Thread(target=work, args=(nm,)) for nm in names]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for nm in names:
        rc, want = oracle_for(shufs(nm)).koc_from_rows(rows, 160)
        assert rc == 0
        assert_same(out[nm], want, nm)


def test_direct_scan_of_pinned_host_rows(capi, shufs, oracle_for):
    """MK_OPT_DIRECT_HOST: rows in pinned host memory are scanned in place over PCIe (no staging copy); rows in ordinary
    memory still take the staged path on the same engine; tickets then stand for finished scans"""
    import ctypes as C
    eng = capi.Engine(shufs("L1K7"), 0)
    try:
        eng.set_option(capi.MK_OPT_DIRECT_HOST, 1)
        rs = np.random.RandomState(77)
        seqs = ui.pool_reads(rs, 30000, 3000) + ui.ragged_reads(rs, 500)
        stride = 304
        rows = ui.rows_from_seqs(seqs, stride)
        n = len(seqs)
        p = C.c_void_p()
        assert capi.lib.mk_host_alloc(C.byref(p), rows.size) == 0
        C.memmove(p, rows.ctypes.data, rows.size)
        eng.begin(capi.MK_MODE_KOC)
        cut1, cut2 = 1000, 2200
        t = C.c_uint64(0)
        capi._check(capi.lib.mk_sketch_push_reads_async(eng.h, p, stride, cut1, 0, C.byref(t)), eng.h)          # pinned: in place
        eng.push_reads(rows[cut1 * stride:cut2 * stride], stride, cut1)                                       # pageable: staged
        t2 = C.c_uint64(0)
        capi._check(capi.lib.mk_sketch_push_reads_async(eng.h, C.c_void_p(p.value + cut2 * stride), stride, n - cut2, cut2, C.byref(t2)), eng.h)
        eng.push_wait(t.value)
        eng.push_wait(t2.value)
        got = eng.finish()
        capi.lib.mk_host_free(p)
    finally:
        eng.close()
    rc, want = oracle_for(shufs("L1K7")).koc_from_rows(rows, stride)
    assert rc == 0
    assert_same(got, want)
