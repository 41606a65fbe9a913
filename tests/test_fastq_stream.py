"""the whole-file FASTQ stream (mk_fastq_stream, metakssd_amd/csrc/host/mk_fastq_stream.c) on the CPU: whatever the file
contains and however it is cut into chunks, the rows it hands on -- and their order and ordinals -- are those of the serial
framers, which tests/test_host.py and tools/fuzz_framing.py pin on the reference's readers (iseq2comem.c:672-673, :343-363)"""
import numpy as np
import pytest

import util_inputs as ui


@pytest.fixture(scope="module")
def capi():
    from metakssd_amd import capi as c
    return c


def seqs_of_rows(rows, stride, n):
    """row payloads up to and including the terminating newline (the zero padding behind it carries nothing)"""
    out = []
    r = rows.reshape(n, stride) if n else np.zeros((0, stride), np.uint8)
    for i in range(n):
        row = r[i].tobytes()
        j = row.find(b"\n")
        out.append(row if j < 0 else row[: j + 1])
        assert j < 0 or not any(row[j + 1:]), "padding behind the newline must be zero"
    return out


def serial_rows(capi, data, occ, TL=14, qmin=54):
    """the serial framer at the widest stride: the definition of what the stream has to deliver"""
    if occ:
        rows, n, nrec, used, rc = capi.fastq_frame_q(data, 4096, TL, qmin=qmin)
    else:
        rows, n, used, rc = capi.fastq_frame(data, 4096)
        nrec = n
    return seqs_of_rows(rows, 4096, n), nrec, rc


def stream_rows(capi, data, occ, T, chunk, TL=14, qmin=54, first=1000, via_fd=False):
    pushes, st, rc = capi.fastq_stream(data, nthreads=T, chunk_bytes=chunk, occ=occ, TL=TL, qmin=qmin, first_ordinal=first, via_fd=via_fd)
    out, ord_expect = [], first
    for rows, stride, n, ord0 in pushes:
        assert stride % 16 == 0 and 32 <= stride <= 4096
        assert stride % 128 != 0 or stride == 4096, "a row pitch of k * 128 bytes halves the scan rate (metakssd_hip.h, MK_ROW_PITCH)"
        assert ord0 == ord_expect, "row ordinals must be consecutive in push order"
        ord_expect += n
        out += seqs_of_rows(rows, stride, n)
    assert st.rows == len(out)
    return out, st, rc


def check(capi, data, occ, threads=(1, 3, 8), chunks=(4096, 5000, 65536, 1 << 20), **kw):
    want, nrec, rc = serial_rows(capi, data, occ, **kw)
    assert rc == 0
    for T in threads:
        for chunk in chunks:
            for via_fd in (False, True):  # the text as a mapping, and through a descriptor the framers pread pieces of (mk_fastq_opts.fd)
                if via_fd and not data:
                    continue
                got, st, rc = stream_rows(capi, data, occ, T, chunk, via_fd=via_fd, **kw)
                assert rc == 0, (T, chunk, via_fd)
                assert got == want, (T, chunk, via_fd, len(got), len(want))
                assert st.records == nrec, (T, chunk, via_fd)
    return want


@pytest.mark.parametrize("occ", [False, True])
@pytest.mark.parametrize("variant", ["plain", "ragged", "crlf", "trunc", "nonl", "long_headers", "short"])
def test_stream_equals_serial_framer(capi, variant, occ):
    rs = np.random.RandomState(5)
    if variant == "plain":
        seqs = [ui.rand_seq(rs, 150) for _ in range(3000)]
    elif variant == "long_headers":
        seqs = [ui.rand_seq(rs, 40) for _ in range(4000)]
    elif variant == "short":
        seqs = [ui.rand_seq(rs, int(rs.randint(0, 30))) for _ in range(6000)]
    else:
        seqs = ui.ragged_reads(rs, 4000)
    quals = ui.random_quals(rs, seqs) if occ else None
    data = ui.fastq_bytes(seqs, crlf=variant == "crlf", final_newline=variant != "nonl", drop_last_qual=variant == "trunc", quals=quals)
    if variant == "long_headers":
        data = data.replace(b"@r", b"@" + b"x" * 700 + b"r")
    want = check(capi, data, occ)
    assert len(want) >= len(seqs) - 1


@pytest.mark.parametrize("occ", [False, True])
def test_stream_through_a_descriptor_in_small_pieces(capi, occ, monkeypatch):
    """mk_fastq_opts.fd: the framers pread PIECES of the file into a buffer of their own and frame each piece's complete records; a
    record cut by a piece's end is framed from the next piece.  With pieces of 84 100 bytes (MK_FS_PIECE; the product reads 1 MiB) every
    chunk of these files is many pieces, records of all lengths up to the readers' line widths straddle their ends, and the rows
    are still the serial framer's"""
    monkeypatch.setenv("MK_FS_PIECE", "84100")
    rs = np.random.RandomState(91)
    seqs = ui.ragged_reads(rs, 6000) + [ui.rand_seq(rs, 150) for _ in range(4000)]
    seqs += [ui.rand_seq(rs, 4000 if not occ else 19000) for _ in range(12)]       # near the readers' fgets widths
    rs.shuffle(seqs)
    quals = ui.random_quals(rs, seqs) if occ else None
    data = ui.fastq_bytes(seqs, quals=quals)
    assert len(data) > 20 * 84100
    want = check(capi, data, occ, threads=(1, 5), chunks=(200000, 1 << 20, 1 << 22))
    assert len(want) >= len(seqs) - 1
    # mk_fastq_opts.early_chunks: only the first chunks are framed before the sink has taken a buffer -- the same rows; also when every
    # early chunk comes to nothing (a file that starts with 300 KB of blank lines: no record start in the first chunks)
    for blob in (data, b"\n" * 300001 + data):
        w2, nrec2, rc = serial_rows(capi, blob, occ)
        assert rc == 0
        for early in (1, 3):
            pushes, st, rc = capi.fastq_stream(blob, nthreads=4, chunk_bytes=100000, occ=occ, TL=14, qmin=54, first_ordinal=0, early_chunks=early, via_fd=True)
            assert rc == 0
            got = [x for rows, stride, n, ord0 in pushes for x in seqs_of_rows(rows, stride, n)]
            assert got == w2 and st.records == nrec2, (early, len(got), len(w2))


@pytest.mark.parametrize("occ", [False, True])
def test_stream_survives_text_built_to_fool_the_boundary_guess(capi, occ):
    """quality lines that start with '@', sequence lines that start with '+' or '@', stray blank lines and missing lines:
    the guessed record starts are then wrong in places, the stream has to notice (chunks discarded / framed serially) and
    still deliver exactly the serial reader's rows"""
    rs = np.random.RandomState(17)
    lines = []
    for r in range(5000):
        L = int(rs.choice([20, 60, 150]))
        s = ui.rand_seq(rs, L)
        q = bytes(rs.choice(np.frombuffer(b"@+#5I", np.uint8), size=L).astype(np.uint8))
        if rs.rand() < 0.3:
            q = b"@" + q[1:]
        hdr = b"@r%d" % r
        if rs.rand() < 0.05:
            s = b"+" + s[1:]
        if rs.rand() < 0.05:
            s = b"@" + s[1:]
        rec = [hdr, s, b"+" if rs.rand() < 0.7 else b"+r%d" % r, q]
        if rs.rand() < 0.01:
            rec.insert(int(rs.randint(0, 5)), b"")  # shifts every later record by one line
        if rs.rand() < 0.01:
            del rec[int(rs.randint(0, len(rec)))]
        lines += rec
    data = b"\n".join(lines) + b"\n"
    want, nrec, rc = serial_rows(capi, data, occ)
    assert rc == 0 and len(want) > 1000
    saw_fallback = False
    for T in (1, 4, 8):
        for chunk in (4096, 20000, 1 << 18):
            got, st, rc = stream_rows(capi, data, occ, T, chunk)
            assert rc == 0 and got == want, (T, chunk)
            saw_fallback |= st.chunks_discarded > 0 or st.serial_rows > 0
    assert saw_fallback, "this input is meant to defeat the guess at least once"


def test_stream_edge_inputs(capi):
    for occ in (False, True):
        for data in (b"", b"\n", b"@r0\n", b"@r0\nACGT\n", b"@r0\nACGT\n+\n", b"@r0\nACGT\n+\nIIII", b"@r0\nACGT\n+\nIIII\n",
                     b"\n\n\n\n\n\n\n\n", b"@r0\nACGT\n+\nIIII\n@r1\nAC", b"ACGT" * 3000, b"@" * 9000 + b"\n" + b"+\n" * 40):
            want, nrec, want_rc = serial_rows(capi, data, occ)
            assert want_rc in (0, capi.MK_ERR_FORMAT)  # the 9000- and 12000-character lines are beyond mt_shortreads2koc's reader
            for T in (1, 4):
                for chunk in (4096, 1 << 20):
                    got, st, rc = stream_rows(capi, data, occ, T, chunk)
                    assert rc == want_rc and (rc != 0 or got == want), (occ, data[:30], T, chunk)


def test_stream_widens_rows_for_longer_reads_and_refuses_overlong_lines(capi):
    rs = np.random.RandomState(3)
    # read lengths grow along the file: the stride sampled at a chunk's head is too small further down
    seqs = [ui.rand_seq(rs, 30 + (i // 40)) for i in range(8000)]
    data = ui.fastq_bytes(seqs)
    check(capi, data, False, threads=(4,), chunks=(1 << 16, 1 << 20))
    # a 4095-character line is beyond mt_shortreads2koc's fgets() width: MK_ERR_FORMAT, like the serial framer
    bad = ui.fastq_bytes([b"ACGT" * 10] * 50 + [b"A" * 4095] + [b"ACGT" * 10] * 50)
    assert capi.fastq_frame(bad, 4096)[3] == capi.MK_ERR_FORMAT
    for chunk in (4096, 1 << 20):
        got, st, rc = stream_rows(capi, bad, False, 4, chunk)
        assert rc == capi.MK_ERR_FORMAT
    # fastq2co's reader cuts reads above 4095 bases into overlapping 4096-byte rows; 19999+ characters are refused
    long_seqs = [ui.rand_seq(rs, int(L)) for L in (100, 5000, 150, 12000, 4095, 4096, 30)] * 6
    datal = ui.fastq_bytes(long_seqs, quals=ui.random_quals(rs, long_seqs))
    check(capi, datal, True, threads=(3,), chunks=(4096, 1 << 16, 1 << 20))
    assert stream_rows(capi, ui.fastq_bytes([b"A" * 19999]), True, 2, 4096)[2] == capi.MK_ERR_FORMAT


def test_threaded_synthetic_fastq_writer_equals_serial(capi, tmp_path):
    for first, n in ((0, 1), (0, 12345), (95, 1200), (999990, 25), (10 ** 9 - 3, 10)):
        a, b = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
        assert capi.lib.mk_synth_fastq_write(a.encode(), 7, first, n, 150) == 0
        for T in (1, 3, 16):
            assert capi.lib.mk_synth_fastq_write_mt(b.encode(), 7, first, n, 150, T) == 0
            assert open(a, "rb").read() == open(b, "rb").read(), (first, n, T)


def test_stream_over_a_file_mapping_with_pages_dropped_behind_the_framers(capi, tmp_path):
    """the command line's configuration: the text is a private read-only mapping of the file and every framer hands its
    chunk's pages back (MADV_DONTNEED) as it goes; re-reads fault back in from the page cache, the rows stay the same"""
    import mmap
    rs = np.random.RandomState(23)
    seqs = ui.ragged_reads(rs, 20000)
    data = ui.fastq_bytes(seqs)
    path = tmp_path / "in.fq"
    path.write_bytes(data)
    want, nrec, rc = serial_rows(capi, data, False)
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, flags=mmap.MAP_PRIVATE, prot=mmap.PROT_READ)
        for T, chunk in ((4, 1 << 18), (8, 1 << 20), (2, 300000)):
            pushes, st, rc = capi.fastq_stream(mm, nthreads=T, chunk_bytes=chunk, drop_pages=True)
            got = []
            for rows, stride, n, ord0 in pushes:
                got += seqs_of_rows(rows, stride, n)
            assert rc == 0 and got == want, (T, chunk)
        assert bytes(mm[:100]) == data[:100]  # a file mapping reads back the file, not zeros
        mm.close()


@pytest.mark.parametrize("length,pitch", [(110, 112), (111, 112), (112, 144), (125, 144), (127, 144), (250, 272), (255, 272), (256, 272), (500, 528)])
def test_stream_row_pitch_avoids_multiples_of_128(capi, length, pitch):
    """reads whose natural pitch (length + newline, rounded up to 16) is 128, 256, 512 get 16 bytes more: every row of a tile
    would otherwise start in the same few L2 channels (profiles/r02_c_probe_read_length.json)"""
    rs = np.random.RandomState(length)
    seqs = [ui.rand_seq(rs, length) for _ in range(300)]
    data = ui.fastq_bytes(seqs)
    pushes, st, rc = capi.fastq_stream(data, nthreads=2, chunk_bytes=1 << 20, occ=0, TL=14, qmin=54, first_ordinal=0)
    assert rc == 0 and st.rows == 300
    assert {stride for _, stride, _, _ in pushes} == {pitch}


# ---- packed rows out of the stream (mk_fastq_opts.packed) -----------------------------------------------------------------------
def _bases_of_text(seq):
    """a sequence line -> [(valid, code)]: what the scan kernel makes of its bytes ((byte >> 1) & 3, valid <=> ACGTacgt)"""
    out = []
    for b in seq:
        code = (b >> 1) & 3
        ok = (b & 0xDF) == b"ACTG"[code]
        out.append((1, code) if ok else (0, 0))
    return out


def _bases_of_packed(row):
    r = np.frombuffer(bytes(row), dtype=np.uint32)
    nb = int(r[0] & 0xFFFF)
    vb = bytes(row)[44:64]
    out = []
    for i in range(nb):
        code = int(r[1 + i // 16] >> (30 - 2 * (i % 16))) & 3
        ok = (vb[i // 8] >> (i % 8)) & 1
        out.append((ok, code))
    allv = int(r[0] >> 16) & 1
    assert allv == int(all(v for v, _ in out)), "the row's all-valid flag"
    return out


@pytest.mark.parametrize("occ", [False, True])
def test_stream_with_a_budgeted_pool_equals_serial_framer(capi, occ):
    """mk_fastq_opts.pool_bytes: buffers sized for PACKED rows from the file's first records, as many as the budget holds (one per chunk
    at most), handed out in address order, every finished buffer reported to the sink (mk_rows_sink.ready) -- rows, order and ordinals
    stay the serial framer's: with room for every chunk, with a budget of a few buffers (reuse), and on a file whose LATER records are
    much shorter than the first ones (more rows a chunk than its buffer holds: those chunks end early, the calling thread frames the rest)"""
    rs = np.random.RandomState(83)
    uniform = [ui.rand_seq(rs, 150) for _ in range(6000)]
    shrinking = [ui.rand_seq(rs, 150) for _ in range(1500)] + [ui.rand_seq(rs, int(rs.randint(1, 12))) for _ in range(20000)]
    for name, seqs in (("uniform", uniform), ("shrinking", shrinking)):
        data = ui.fastq_bytes(seqs, quals=[bytes(rs.randint(44, 64, len(x)).astype(np.uint8)) for x in seqs])
        want, nrec, rc = serial_rows(capi, data, occ)
        assert rc == 0
        want_bases = [_bases_of_text(w[:-1] if w.endswith(b"\n") else w) for w in want]
        for T, chunk, pool in ((4, 1 << 16, 1 << 30), (3, 20000, 200000), (8, 1 << 15, 1 << 30), (1, 1 << 20, 1 << 22)):
            log = []
            pushes, st, rc = capi.fastq_stream(data, nthreads=T, chunk_bytes=chunk, occ=occ, TL=14, qmin=54, first_ordinal=3, packed=True, pool_bytes=pool,
                                               ready_log=log)
            assert rc == 0, (name, T, chunk, pool)
            got, ord_expect = [], 3
            for rows, stride, n, ord0 in pushes:
                assert ord0 == ord_expect
                ord_expect += n
                if stride & capi.MK_ROWS_PACKED:
                    got += [_bases_of_packed(rows[64 * i: 64 * i + 64]) for i in range(n)]
                else:
                    got += [_bases_of_text(x[:-1] if x.endswith(b"\n") else x) for x in seqs_of_rows(rows, stride, n)]
            assert got == want_bases, (name, T, chunk, pool)
            assert st.records == nrec
            blocks, ready = log[0], log[1:]
            assert len(blocks) == 1 and len(ready) >= st.chunks - st.chunks_discarded - 1 - (1 if st.serial_rows else 0) or st.serial_rows
            assert all(0 <= off and off + n <= blocks[0][1] for off, n in ready), "every reported buffer lies inside the sink's block"
            if name == "uniform" and pool == 1 << 30:
                nchunks = (len(data) + chunk - 1) // chunk
                assert len({off for off, n in ready}) <= nchunks  # (a buffer that has come back is taken again before a fresh one)
                offs = [off for off, n in ready]
                assert len(set(n for off, n in ready)) == 1 and min(o_ for o_ in offs if o_) == ready[0][1], "buffer 0 is the serial fallback's; the framers' follow it"
            if name == "uniform" and chunk == 1 << 20:
                # packed sizing: a buffer holds about a fifth of its chunk's text (text rows: all of it and an eighth more)
                assert ready[0][1] < 0.3 * chunk, ready[0][1]
            if name == "shrinking":
                assert st.serial_rows > 0 or pool < (1 << 23)


@pytest.mark.parametrize("occ", [False, True])
def test_stream_packed_rows_carry_the_serial_framers_bases(capi, occ):
    """mk_fastq_opts.packed: buffers whose reads all fit 152 bases come as 64-byte packed rows -- base for base (code, validity) what the
    serial framer's text rows hold; a buffer with a longer read comes as text rows; order and ordinals as ever"""
    rs = np.random.RandomState(81)
    seqs = []
    for i in range(3000):
        n = int(rs.randint(1, 153)) if i % 50 else int(rs.randint(0, 3))
        s = bytearray(ui.rand_seq(rs, n))
        if n and i % 7 == 0:
            s[int(rs.randint(0, n))] = ord("N")
        if i % 11 == 0:
            s = bytearray(bytes(s).lower())
        seqs.append(bytes(s))
    seqs[1700] = ui.rand_seq(rs, 400)          # one long read: its buffer falls back to text rows
    data = ui.fastq_bytes(seqs, quals=[bytes(rs.randint(44, 64, len(x)).astype(np.uint8)) for x in seqs])
    want, nrec, rc = serial_rows(capi, data, occ)
    assert rc == 0
    want_bases = [_bases_of_text(w[:-1] if w.endswith(b"\n") else w) for w in want]
    for T, chunk in ((1, 1 << 20), (3, 20000), (8, 4096 * 5)):
        pushes, st, rc = capi.fastq_stream(data, nthreads=T, chunk_bytes=chunk, occ=occ, TL=14, qmin=54, first_ordinal=7, packed=True)
        assert rc == 0
        got, ord_expect, n_packed = [], 7, 0
        for rows, stride, n, ord0 in pushes:
            assert ord0 == ord_expect
            ord_expect += n
            if stride & capi.MK_ROWS_PACKED:
                assert stride == (64 | capi.MK_ROWS_PACKED)
                n_packed += n
                got += [_bases_of_packed(rows[64 * i: 64 * i + 64]) for i in range(n)]
            else:
                got += [_bases_of_text(x[:-1] if x.endswith(b"\n") else x) for x in seqs_of_rows(rows, stride, n)]
        assert st.records == nrec
        assert got == want_bases, (T, chunk)
        assert n_packed > 0 and (n_packed < len(got) or T == 1 and chunk >= len(data))
