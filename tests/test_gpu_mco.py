"""stage II (inverted index) and the `dist -r` shared-k-mer counting on the device vs the oracle:
bit-exact gid lists, row table, dense-index slabs and count matrices."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mco():
    from metakssd_amd import capi
    m = capi.Mco(0)
    yield m
    m.close()


def sketches(rs, nsk, universe, lo, hi, empty=(), extremes=False):
    """nsk id sets drawn from a common universe (so that rows are shared), some empty; -> (ids, index)"""
    pool = np.unique(rs.randint(0, 2 ** 32, size=universe, dtype=np.uint64).astype(np.uint32))
    if extremes:
        pool = np.unique(np.concatenate([pool, np.array([0, 1, 2 ** 32 - 1, 2 ** 32 - 2, 2 ** 27, 2 ** 27 - 1], np.uint32)]))
    parts, index = [], [0]
    for j in range(nsk):
        n = 0 if j in empty else int(rs.randint(lo, hi + 1))
        p = rs.permutation(pool)[:min(n, pool.size)]     # hash-table order in the real files: unsorted
        parts.append(p)
        index.append(index[-1] + p.size)
    ids = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
    return ids.astype(np.uint32), np.array(index, np.uint64)


@pytest.mark.parametrize("nsk,universe,lo,hi,empty,extremes", [
    (4, 300, 50, 200, (), True),
    (1, 50, 50, 50, (), False),
    (7, 2000, 0, 1500, (0, 3, 6), True),
    (300, 20000, 100, 3000, (5,), False),
    (3, 10, 0, 0, (0, 1, 2), False),
    (40, 400000, 50000, 150000, (), True),
])
def test_build_matches_oracle(mco, nsk, universe, lo, hi, empty, extremes):
    rs = np.random.RandomState(nsk * 7 + universe)
    ids, index = sketches(rs, nsk, universe, lo, hi, empty, extremes)
    g, ri, re_ = mco.build(ids, index)
    og, ori, ore = ob.mco_build(ids, index)
    assert np.array_equal(g, og)
    assert np.array_equal(ri, ori) and np.array_equal(re_, ore)
    # dense index: the first slab, the last rows, and rows around table entries
    def want(row0, nrows):
        rows = np.arange(row0, row0 + nrows, dtype=np.uint64)
        u = np.searchsorted(ori.astype(np.uint64), rows, side="right")
        ends = np.concatenate([[0], ore]).astype(np.uint64)
        return ends[u]
    spots = [(0, 5000), (2 ** 32 - 4096, 4096), (2 ** 27 - 1000, 2000)]
    if ori.size:
        mid = int(ori[ori.size // 2])
        spots.append((max(0, mid - 300), 600 if mid + 300 < 2 ** 32 else 300))
    for row0, nrows in spots:
        assert np.array_equal(mco.index_rows(row0, nrows), want(row0, nrows)), (row0, nrows)


def test_index_full_slab_properties(mco):
    """one whole 2^27-row slab: monotone, starts/ends at the table's values, changes exactly at the rows of the table"""
    rs = np.random.RandomState(5)
    ids, index = sketches(rs, 6, 5000, 500, 3000, (), True)
    _, ri, re_ = mco.build(ids, index)
    slab = mco.index_rows(0, 2 ** 27)
    d = np.flatnonzero(np.diff(slab))
    inslab = ri[ri < 2 ** 27]
    first_is_row = int(inslab.size and inslab[0] == 0)
    assert np.array_equal(d + 1, inslab[first_is_row:].astype(np.int64))
    assert slab[-1] == (re_[inslab.size - 1] if inslab.size else 0)


@pytest.mark.parametrize("nref,nqry,universe,lo,hi,zero_ct", [
    (4, 3, 400, 50, 300, (2,)),            # LDS counters
    (900, 20, 30000, 500, 8000, ()),       # LDS, long rows (cooperative walk)
    (40000, 6, 200000, 1, 6, (1,)),        # more references than LDS counters: global atomics
    (70000, 4, 300000, 1, 5, ()),          # more than 65 535 references: 32-bit genome lists
    (3000, 400, 20000, 5, 40, ()),         # many tiny query sketches: global atomics by the slice heuristic
    (2, 1, 100000, 60000, 90000, ()),      # one big query sketch cut into slices
])
def test_count_matches_oracle(mco, nref, nqry, universe, lo, hi, zero_ct):
    rs = np.random.RandomState(nref + nqry)
    pool = np.unique(rs.randint(0, 2 ** 32, size=universe, dtype=np.uint64).astype(np.uint32))
    def draw(nsk):
        parts, index = [], [0]
        for _ in range(nsk):
            n = int(rs.randint(lo, hi + 1))
            if n * 8 < pool.size:     # small sketch: distinct random picks without shuffling the whole pool
                p = pool[np.unique(rs.randint(0, pool.size, size=n))]
                p = p[rs.permutation(p.size)]
            else:
                p = rs.permutation(pool)[:n]
            parts.append(p)
            index.append(index[-1] + p.size)
        return np.concatenate(parts).astype(np.uint32), np.array(index, np.uint64)
    rids, rindex = draw(nref)
    qids, qindex = draw(nqry)
    qids[::17] ^= 1                      # some ids that are in no row
    ctx = np.diff(qindex).astype(np.uint32)
    for k in zero_ct:
        ctx[k] = 0                       # stat says empty: skipped whatever the lists hold (command_dist.c:1035)
    og, ori, ore = ob.mco_build(rids, rindex)
    want = ob.mco_count(og, ori, ore, qids, qindex, ctx, nref)
    g, ri, re_ = mco.build(rids, rindex)
    got = mco.count(nref, qindex, ctx, [{"qry_ids": qids}])               # device row table
    assert np.array_equal(got, want)
    # extents as the CLI takes them from the mmap'ed dense index
    u = np.searchsorted(ori, qids, side="left")
    hit = (u < ori.size) & (ori[np.minimum(u, ori.size - 1)] == qids)
    ends = np.concatenate([[0], ore]).astype(np.uint64)
    es = np.where(hit, ends[np.minimum(u, ori.size - 1)], 0).astype(np.uint64)
    ee = np.where(hit, ends[np.minimum(u, ori.size - 1) + 1], 0).astype(np.uint64)
    got2 = mco.count(nref, qindex, ctx, [{"gids": og, "ext_start": es, "ext_end": ee}])
    assert np.array_equal(got2, want)
    # two components accumulate
    got3 = mco.count(nref, qindex, ctx, [{"gids": og, "ext_start": es, "ext_end": ee}, {"gids": og, "ext_start": es, "ext_end": ee}])
    assert np.array_equal(got3, want * 2)
    assert want.sum() > 0


def test_count_properties_large(mco):
    """size-independent checks at a size the oracle does not see: row sums = total row lengths of the query's ids;
    a reference searched against itself has its sketch size on the diagonal"""
    rs = np.random.RandomState(77)
    nref = 2000
    pool = np.unique(rs.randint(0, 2 ** 32, size=3_000_000, dtype=np.uint64).astype(np.uint32))
    sizes = rs.randint(20000, 60000, size=nref)
    starts = rs.randint(0, pool.size - 60000, size=nref)
    parts = [pool[s:s + n] for s, n in zip(starts, sizes)]           # overlapping windows: heavy sharing
    ids = np.concatenate(parts)
    index = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    g, ri, re_ = mco.build(ids, index)
    assert g.size == ids.size and np.all(np.diff(ri.astype(np.int64)) > 0) and re_[-1] == ids.size
    sel = [3, 500, 1999]
    qids = np.concatenate([parts[i] for i in sel])
    qindex = np.concatenate([[0], np.cumsum([parts[i].size for i in sel])]).astype(np.uint64)
    ct = mco.count(nref, qindex, np.diff(qindex).astype(np.uint32), [{"qry_ids": qids}])
    lens = np.diff(np.concatenate([[0], re_]).astype(np.int64))
    for row, i in enumerate(sel):
        assert ct[row, i] == parts[i].size
        u = np.searchsorted(ri, parts[i])
        assert int(ct[row].sum()) == int(lens[u].sum())
        assert np.all(ct[row] <= np.minimum(parts[i].size, sizes))


def test_state_and_argument_errors(mco):
    from metakssd_amd import capi
    m = capi.Mco(0)
    with pytest.raises(capi.MkError):
        m.index_rows(0, 16)                                   # before any build
    with pytest.raises(capi.MkError):
        m.build(np.arange(4, dtype=np.uint32), np.array([0, 3, 2, 4], np.uint64))   # index not ascending
    with pytest.raises(capi.MkError) as big:                   # 2^32 ids and more: the sort's 32-bit prefixes would wrap
        m.build(np.arange(4, dtype=np.uint32), np.array([0, 2 ** 32], np.uint64))
    assert big.value.code == capi.MK_ERR_ARG and "2^32" in str(big.value)
    m.build(np.arange(4, dtype=np.uint32), np.array([0, 2, 4], np.uint64))
    with pytest.raises(capi.MkError):
        m.index_rows(2 ** 32 - 8, 16)                         # past the last row
    with pytest.raises(capi.MkError):                          # extents beyond the gid lists
        m.count(2, np.array([0, 1], np.uint64), [1], [{"gids": np.zeros(3, np.uint32), "ext_start": np.array([2], np.uint64),
                                                       "ext_end": np.array([9], np.uint64)}])
    m.close()


@pytest.mark.parametrize("opts", [{2: 1}, {1: 1}, {1: 1, 2: 1}])
def test_count_kernel_variants_agree(mco, opts):
    """mk_mco_set_option selects the other instantiations of the counting kernel (MK_MCO_OPT_WIDE_LISTS = 2: 32-bit lists with
    LDS counters, MK_MCO_OPT_GLOBAL_COUNTERS = 1: global atomics on a small database): same matrix"""
    rs = np.random.RandomState(17)
    pool = np.unique(rs.randint(0, 2 ** 32, size=40000, dtype=np.uint64).astype(np.uint32))
    def draw(nsk, lo, hi):
        parts, index = [], [0]
        for _ in range(nsk):
            p = rs.permutation(pool)[:int(rs.randint(lo, hi + 1))]
            parts.append(p)
            index.append(index[-1] + p.size)
        return np.concatenate(parts).astype(np.uint32), np.array(index, np.uint64)
    rids, rindex = draw(700, 200, 9000)
    qids, qindex = draw(9, 0, 20000)
    ctx = np.diff(qindex).astype(np.uint32)
    og, ori, ore = ob.mco_build(rids, rindex)
    want = ob.mco_count(og, ori, ore, qids, qindex, ctx, 700)
    mco.build(rids, rindex)
    from metakssd_amd import capi
    try:
        for k, v in opts.items():
            assert capi.lib.mk_mco_set_option(mco.h, k, v) == 0
        got = mco.count(700, qindex, ctx, [{"qry_ids": qids}])
    finally:
        for k in (1, 2):
            capi.lib.mk_mco_set_option(mco.h, k, 0)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1023, 1024, 4095, 4096, 4097, 8191, 8192, 8193, 16385, 100003, 3000017])
def test_radix_sort_pairs_is_stable_and_sorted(n):
    """the hand-written LSD radix sort behind mk_mco_build (mk_sort.hip.h): against numpy's stable argsort on keys drawn so that
    passes are skipped (few distinct top bytes), ties are long (few distinct keys), and every digit value occurs"""
    from metakssd_amd import capi
    rs = np.random.RandomState(1000 + n % 97)
    mco = capi.Mco(0)
    try:
        for kind in ("uniform", "small range", "few keys", "one key", "top byte only", "descending"):
            if kind == "uniform":
                k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
            elif kind == "small range":
                k = rs.randint(0, 70000, size=n).astype(np.uint32)
            elif kind == "few keys":
                k = rs.choice(np.array([0, 1, 255, 256, 65535, 65536, 2 ** 24, 2 ** 32 - 1], dtype=np.uint64), size=n).astype(np.uint32)
            elif kind == "one key":
                k = np.full(n, 123456789, np.uint32)
            elif kind == "top byte only":
                k = (rs.randint(0, 256, size=n).astype(np.uint64) << 24).astype(np.uint32)
            else:
                k = np.arange(n, 0, -1, dtype=np.uint64).astype(np.uint32) * np.uint32(3)
            v = np.arange(n, dtype=np.uint32)  # the value is the input position: stability is visible in it
            gk, gv = mco.sort_pairs(k, v)
            order = np.argsort(k, kind="stable")
            assert np.array_equal(gk, k[order]), (n, kind)
            assert np.array_equal(gv, v[order]), (n, kind)
    finally:
        mco.close()


def test_build_large_pageable_input_and_views(mco):
    """ids that cross several 64 MiB staging pieces of the threaded upload (mk_mco_upload), twice in a row on one handle (the
    second call reuses staging buffers the first one's copies went through); copy=False hands out the library's pinned buffers"""
    rs = np.random.RandomState(77)
    n = 40 * 1000 * 1000 + 12345                  # 160 MB of ids: two full pieces and a ragged third
    nsk = 500
    cuts = np.sort(rs.randint(0, n, size=nsk - 1))
    index = np.concatenate([[0], cuts, [n]]).astype(np.uint64)
    for rep in range(2):
        ids = rs.randint(0, 2 ** 24, size=n, dtype=np.uint64).astype(np.uint32)   # 16 M rows: long lists
        og, ori, ore = ob.mco_build(ids, index)
        g, ri, re_ = mco.build(ids, index, copy=False)
        assert g.size == n and np.array_equal(g, og), rep
        assert np.array_equal(ri, ori) and np.array_equal(re_, ore), rep
    g2, _, _ = mco.build(ids[:1000], np.array([0, 400, 1000], np.uint64))       # views of the last build are gone by contract
    assert g2.size == 1000


def test_index_rows_pinned_equals_pageable(mco):
    rs = np.random.RandomState(3)
    ids, index = sketches(rs, 50, 100000, 1000, 20000)
    mco.build(ids, index)
    for row0, nrows in ((0, 1 << 16), (2 ** 31, 1 << 20), (2 ** 32 - 4096, 4096)):
        assert np.array_equal(mco.index_rows(row0, nrows, pinned=True), mco.index_rows(row0, nrows))
