"""GPU parity for `set -u` / `set -q`: mk_setop_* through the C ABI against the oracle's restatement
of sketch_union() / uniq_sketch_union() (command_set.c:241-319, 427-512) and against numpy's definition of the same
sets.  The oracle side is pinned by tests/golden (pan.N / uniq_pan.N written by the compiled reference)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from metakssd_amd import capi as c
    if c.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests must run on the MI355X box")
    return c


@pytest.fixture(scope="module")
def setop(capi):
    s = capi.SetOp(0)
    yield s
    s.close()


def oracle_union(ids, uniq):
    from oracle_binding import load
    lib = load()
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    out = np.zeros(max(1, ids.size), np.uint32)
    lib.ko_set_union.restype = C.c_size_t
    lib.ko_set_union.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    m = lib.ko_set_union(ids.ctypes.data if ids.size else None, ids.size, 1 if uniq else 0, out.ctypes.data)
    return out[:m]


def numpy_union(ids, uniq):
    vals, cnt = np.unique(np.asarray(ids, dtype=np.uint32), return_counts=True)
    return vals[cnt == 1] if uniq else vals


@pytest.mark.parametrize("uniq", [False, True])
def test_union_of_sketch_like_lists(capi, setop, uniq):
    """three 'genomes' sharing part of their ids, handed over as separate lists (what combco.N concatenates)"""
    rs = np.random.RandomState(61)
    pool = rs.randint(0, 2 ** 32, size=300000, dtype=np.uint64).astype(np.uint32)
    lists = [rs.permutation(pool[:200000]), rs.permutation(pool[100000:]), rs.permutation(pool[50000:250000])]
    got = setop.union(lists, uniq=uniq)
    allids = np.concatenate(lists)
    want = oracle_union(allids, uniq)
    assert np.array_equal(got, want)
    assert np.array_equal(want, numpy_union(allids, uniq))
    assert got.size > 0


@pytest.mark.parametrize("uniq", [False, True])
def test_word_and_range_edges(capi, setop, uniq):
    edge = np.array([0, 1, 31, 32, 33, 63, 64, 65, 1023, 1024, 32767, 32768, 32769, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 2,
                     2 ** 32 - 1, 0, 64, 2 ** 32 - 1, 2 ** 32 - 1], dtype=np.uint64).astype(np.uint32)
    got = setop.union([edge], uniq=uniq)
    assert np.array_equal(got, oracle_union(edge, uniq))
    assert np.array_equal(got, numpy_union(edge, uniq))


def test_empty_input_and_handle_reuse(capi, setop):
    assert setop.union([], uniq=False).size == 0
    assert setop.union([np.zeros(0, np.uint32)], uniq=True).size == 0
    a = np.arange(1000, 2000, dtype=np.uint32)
    assert np.array_equal(setop.union([a, a], uniq=False), a)
    assert setop.union([a, a], uniq=True).size == 0          # everything twice
    assert np.array_equal(setop.union([a], uniq=True), a)     # begin() cleared both dictionaries


def test_dense_run_and_unaligned_list(capi, setop):
    """a dense range (every bit of many words set) and a list whose address is not 16-byte aligned"""
    base = np.arange(5_000_000, 9_000_000, dtype=np.uint32)
    buf = np.concatenate([np.zeros(1, np.uint32), base[::-1], base[::7]])
    view = buf[1:]                       # 4 bytes past an aligned address
    assert view.ctypes.data % 16 == 4
    got = setop.union([view], uniq=False)
    assert np.array_equal(got, base)
    got = setop.union([view], uniq=True)
    mask = np.ones(base.size, bool)
    mask[::7] = False
    assert np.array_equal(got, base[mask])


def test_large_random_list_properties(capi, setop):
    """40 M ids (160 MB: several staging buffers): ascending, duplicate-free, same set as numpy's"""
    rs = np.random.RandomState(62)
    ids = rs.randint(0, 2 ** 28, size=40_000_000, dtype=np.int64).astype(np.uint32)
    got = setop.union([ids[:15_000_000], ids[15_000_000:]], uniq=False)
    assert (np.diff(got.astype(np.int64)) > 0).all()
    assert np.array_equal(got, np.unique(ids))
    gotq = setop.union([ids], uniq=True)
    assert np.array_equal(gotq, numpy_union(ids, True))


def test_device_resident_lists(capi, setop):
    hip = C.CDLL("libamdhip64.so")
    rs = np.random.RandomState(63)
    ids = rs.randint(0, 2 ** 32, size=1_000_000, dtype=np.uint64).astype(np.uint32)
    ids[::10] = ids[5]
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(ids.nbytes)) == 0
    assert hip.hipMemcpy(p, C.c_void_p(ids.ctypes.data), C.c_size_t(ids.nbytes), 1) == 0
    setop.begin(uniq=True)
    setop.add_device(p.value, ids.size // 2)
    setop.add_device(p.value + 4 * (ids.size // 2), ids.size - ids.size // 2)
    n = setop.finish_count()
    hip.hipFree(p)
    assert n == numpy_union(ids, True).size


# ---- set -i / -s: ordered filter by dictionary membership (sketch_operate, command_set.c:392-405) --------------------
def oracle_filter(pan, keep, ids):
    from oracle_binding import load
    lib = load()
    lib.ko_set_filter.restype = C.c_size_t
    lib.ko_set_filter.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    pan = np.ascontiguousarray(pan, np.uint32)
    ids = np.ascontiguousarray(ids, np.uint32)
    out = np.zeros(max(1, ids.size), np.uint32)
    m = lib.ko_set_filter(pan.ctypes.data if pan.size else None, pan.size, keep, ids.ctypes.data if ids.size else None, ids.size,
                          out.ctypes.data)
    return out[:m]


def product_filter(capi, setop, pan, keep, ids, bounds):
    lib = capi.lib
    lib.mk_setop_filter.restype = C.c_int
    lib.mk_setop_filter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_uint64), C.c_void_p]
    pan = np.ascontiguousarray(pan, np.uint32)
    ids = np.ascontiguousarray(ids, np.uint32)
    bounds = np.ascontiguousarray(bounds, np.uint64)
    bout = np.zeros(bounds.size, np.uint64)
    setop.begin(uniq=False)
    if pan.size:
        assert lib.mk_setop_add(setop.h, pan.ctypes.data, pan.size) == 0
    out, n = C.c_void_p(), C.c_uint64(0)
    rc = lib.mk_setop_filter(setop.h, keep, ids.ctypes.data if ids.size else None, ids.size, bounds.ctypes.data if bounds.size else None,
                             bounds.size, C.byref(out), C.byref(n), bout.ctypes.data if bounds.size else None)
    assert rc == 0, lib.mk_setop_last_error(setop.h)
    got = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint32)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
    return got, bout


@pytest.mark.parametrize("keep", [1, 0])
@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 1023, 1024, 1025, 300001])
def test_filter_keeps_order_and_reports_boundaries(capi, setop, keep, n):
    rs = np.random.RandomState(70 + n % 97)
    universe = rs.randint(0, 2 ** 32, size=50000, dtype=np.uint64).astype(np.uint32)
    pan = universe[::2]
    ids = universe[rs.randint(0, universe.size, size=n)] if n else np.zeros(0, np.uint32)   # repeats, arbitrary order
    cuts = np.unique(np.concatenate([[0, n], rs.randint(0, n + 1, size=7)])).astype(np.uint64)
    got, bout = product_filter(capi, setop, pan, keep, ids, cuts)
    want = oracle_filter(pan, keep, ids)
    assert np.array_equal(got, want)
    member = np.isin(ids, pan)
    kept = member if keep else ~member
    assert np.array_equal(want, ids[kept])
    assert np.array_equal(bout, np.concatenate([[0], np.cumsum(kept)])[cuts.astype(np.int64)].astype(np.uint64))


def test_filter_empty_dictionary_and_state_errors(capi, setop):
    ids = np.arange(10, 5000, 3, dtype=np.uint32)
    got, _ = product_filter(capi, setop, np.zeros(0, np.uint32), 0, ids, np.zeros(0, np.uint64))
    assert np.array_equal(got, ids)                      # nothing is a member: -s keeps all
    got, _ = product_filter(capi, setop, np.zeros(0, np.uint32), 1, ids, np.zeros(0, np.uint64))
    assert got.size == 0                                 # ... and -i keeps none
    setop.begin(uniq=True)                               # a uniq dictionary is not a membership dictionary
    out, n = C.c_void_p(), C.c_uint64(0)
    rc = capi.lib.mk_setop_filter(setop.h, 1, ids.ctypes.data, ids.size, None, 0, C.byref(out), C.byref(n), None)
    assert rc == capi.MK_ERR_STATE
    setop.finish_count()


# ---- set -g: the per-taxon FCFS table of grouping_genomes() (command_set.c:866-915) ------------------------------------
def oracle_group(ids, table_size):
    from oracle_binding import load
    lib = load()
    lib.ko_group_layout.restype = C.c_size_t
    lib.ko_group_layout.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
    ids = np.ascontiguousarray(ids, np.uint32)
    out = np.zeros(max(1, ids.size), np.uint32)
    m = lib.ko_group_layout(ids.ctypes.data if ids.size else None, ids.size, table_size, out.ctypes.data)
    return out[:m]


def product_group(capi, setop, ids, table_size):
    lib = capi.lib
    lib.mk_setop_group.restype = C.c_int
    lib.mk_setop_group.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    ids = np.ascontiguousarray(ids, np.uint32)
    out, n = C.c_void_p(), C.c_uint64(0)
    rc = lib.mk_setop_group(setop.h, ids.ctypes.data if ids.size else None, ids.size, table_size, C.byref(out), C.byref(n))
    assert rc == 0, lib.mk_setop_last_error(setop.h)
    return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint32)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)


def table_size_for(capi, n):
    capi.lib.mk_setop_group_table_size.restype = C.c_uint32
    capi.lib.mk_setop_group_table_size.argtypes = [C.c_uint64]
    return capi.lib.mk_setop_group_table_size(n)


def test_group_table_size_matches_reference_formula(capi):
    from oracle_binding import load
    lib = load()
    lib.ko_group_table_size.restype = C.c_uint32
    lib.ko_group_table_size.argtypes = [C.c_uint64]
    for n in [1, 2, 85, 86, 170, 171, 341, 342, 1000, 1365, 1366, 43690, 43691, 10 ** 6, 10 ** 7, 3 * 10 ** 8, 14 * 10 ** 8]:
        assert table_size_for(capi, n) == lib.ko_group_table_size(n), n
    assert table_size_for(capi, 1000) == 2039 and table_size_for(capi, 100) == 251  # primer[LOG2(1.5 n) - 7]


@pytest.mark.parametrize("n,dup", [(1, 1), (17, 2), (1000, 3), (1365, 1), (50000, 4), (400000, 3)])
def test_group_layout_equals_sequential_fcfs_table(capi, setop, n, dup):
    """a taxon of `dup` genomes sharing ids: slot-order dump of the FCFS table == the oracle's sequential insert"""
    rs = np.random.RandomState(80 + n % 89)
    base = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    base[rs.randint(0, n, size=max(1, n // 50))] = 0                      # id 0 is never stored
    parts = [base] + [base[rs.permutation(n)[: max(1, n // 2)]] for _ in range(dup - 1)]
    ids = np.concatenate(parts)
    S = table_size_for(capi, ids.size)
    got = product_group(capi, setop, ids, S)
    want = oracle_group(ids, S)
    assert np.array_equal(got, want)
    assert set(got.tolist()) == set(np.unique(ids).tolist()) - {0}


def test_group_dense_and_overfull_tables(capi, setop):
    """tables the reference would never choose: load 0.97, then more distinct ids than slots (ids without a place after
    table_size probes are dropped, 32-bit probe arithmetic wraps for the long walks) -- still the sequential result"""
    rs = np.random.RandomState(81)
    ids = rs.randint(1, 2 ** 32, size=4000, dtype=np.uint64).astype(np.uint32)
    for S in (4093, 2039, 1021):
        got = product_group(capi, setop, ids, S)
        want = oracle_group(ids, S)
        assert np.array_equal(got, want), S
        assert got.size == min(S, np.unique(ids).size)
    big = rs.randint(1, 2 ** 32, size=133000, dtype=np.uint64).astype(np.uint32)
    got = product_group(capi, setop, big, 131071)
    assert np.array_equal(got, oracle_group(big, 131071))


def test_group_is_independent_of_dictionary_state(capi, setop):
    rs = np.random.RandomState(82)
    ids = rs.randint(0, 2 ** 20, size=30000, dtype=np.uint64).astype(np.uint32)
    S = table_size_for(capi, ids.size)
    a = product_group(capi, setop, ids, S)
    setop.begin(uniq=True)
    b = product_group(capi, setop, ids, S)
    setop.finish_count()
    assert np.array_equal(a, b) and np.array_equal(a, oracle_group(ids, S))
    assert product_group(capi, setop, np.zeros(0, np.uint32), 251).size == 0
    assert product_group(capi, setop, np.zeros(5, np.uint32), 251).size == 0       # only id 0: nothing stored


# ---- composite -q: the join of get_species_abundance() (command_composite.c:525-553) ------------------------------------
def product_join(capi, setop, qids, qab, rids, bounds):
    lib = capi.lib
    lib.mk_setop_join.restype = C.c_int
    lib.mk_setop_join.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    qids = np.ascontiguousarray(qids, np.uint32)
    qab = np.ascontiguousarray(qab, np.uint16)
    rids = np.ascontiguousarray(rids, np.uint32)
    bounds = np.ascontiguousarray(bounds, np.uint64)
    bout = np.zeros(bounds.size, np.uint64)
    out, n = C.c_void_p(), C.c_uint64(0)
    rc = lib.mk_setop_join(setop.h, qids.ctypes.data if qids.size else None, qab.ctypes.data if qab.size else None, qids.size,
                           rids.ctypes.data if rids.size else None, rids.size, bounds.ctypes.data if bounds.size else None, bounds.size,
                           C.byref(out), C.byref(n), bout.ctypes.data if bounds.size else None)
    assert rc == 0, lib.mk_setop_last_error(setop.h)
    got = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint32)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
    return got, bout


@pytest.mark.parametrize("nq,nref", [(0, 100), (100, 0), (1, 1), (1000, 5000), (200000, 1500000)])
def test_join_returns_query_counts_in_reference_order(capi, setop, nq, nref):
    """query = distinct ids with counts (a -A sketch), reference = blocks of ids (ids may repeat across blocks): for every
    reference position whose id is in the query, the query's count, in reference order, with the per-block boundaries"""
    rs = np.random.RandomState(90 + nq % 83)
    universe = np.unique(rs.randint(0, 2 ** 32, size=max(4, 2 * nq), dtype=np.uint64).astype(np.uint32))
    if nq:
        universe[0] = 0                                           # id 0 is an ordinary id here (the dictionary stores idx + 1)
    qids = rs.permutation(universe)[:nq]
    qab = rs.randint(1, 65536, size=nq).astype(np.uint16)
    rids = universe[rs.randint(0, universe.size, size=nref)] if nref else np.zeros(0, np.uint32)
    cuts = np.unique(np.concatenate([[0, nref], rs.randint(0, nref + 1, size=9)])).astype(np.uint64)
    got, bout = product_join(capi, setop, qids, qab, rids, cuts)
    lut = dict(zip(qids.tolist(), qab.tolist()))
    hit = np.array([r in lut for r in rids.tolist()], dtype=bool) if nref else np.zeros(0, bool)
    want = np.array([lut[r] for r in rids[hit].tolist()], dtype=np.uint32)
    assert np.array_equal(got, want)
    assert np.array_equal(bout, np.concatenate([[0], np.cumsum(hit)])[cuts.astype(np.int64)].astype(np.uint64))
    if nq and nref > 1:
        assert got.size > 0


def test_join_duplicate_query_ids_take_the_first_occurrence(capi, setop):
    """not produced by dist (sketch ids are distinct), but defined by the reference's dictionary: the first inserted wins"""
    qids = np.array([7, 9, 7, 11, 9], np.uint32)
    qab = np.array([70, 90, 71, 110, 91], np.uint16)
    got, _ = product_join(capi, setop, qids, qab, np.array([9, 7, 5, 11, 7], np.uint32), np.zeros(0, np.uint64))
    assert got.tolist() == [90, 70, 110, 70]
