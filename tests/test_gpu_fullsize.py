"""BASELINE.json's full size on one GPU (config 3: 50 M synthetic 150 bp reads, L3K11, -A), checked through properties that
do not need the CPU oracle (which would take minutes at this size):

  * the sketch is a deterministic function of the input (two runs: identical bytes)
  * cutting the read stream into several pushes does not change a byte (arrival order / batching independence)
  * two engines sketching the two halves with global ordinals, merged through mk_partial_export/import, equal the single
    engine (the multi-GPU claim of SURVEY.md 8e at full size)
  * counts: every id once, every count >= 1, the number of distinct keys bench.py reports for this workload, the sum of
    counts within 5 sigma of the expected number of accepted k-mers (129 per read x 1/4096)

The same engine is checked against the oracle bit for bit at oracle-sized inputs in test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, LEN, STRIDE, SEED = 50_000_000, 150, 160, 20261002


@pytest.fixture(scope="module")
def capi():
    from metakssd_amd import capi as c
    if c.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests must run on the MI355X box")
    return c


@pytest.fixture(scope="module")
def reads_dev(capi):
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(N * STRIDE)) == 0
    capi.synth_rows_device(0, None, SEED, 0, N, LEN, STRIDE, p.value)
    assert hip.hipDeviceSynchronize() == 0
    yield p.value
    hip.hipFree(p)


def sketch(capi, eng, ptr, pushes):
    eng.begin(capi.MK_MODE_KOC)
    per = (N + pushes - 1) // pushes
    done = 0
    while done < N:
        m = min(per, N - done)
        eng.push_reads_device(ptr + done * STRIDE, STRIDE, m, done)
        done += m
    return eng.finish()[0]


def test_full_size_sketch_properties(capi, shufs, reads_dev):
    shuf = shufs("L3K11")
    hip = C.CDLL("libamdhip64.so")
    e0, e1 = capi.Engine(shuf, 0), capi.Engine(shuf, 0)
    try:
        ids, cnt = sketch(capi, e0, reads_dev, 1)
        ids2, cnt2 = sketch(capi, e0, reads_dev, 1)
        assert np.array_equal(ids, ids2) and np.array_equal(cnt, cnt2)            # deterministic
        ids7, cnt7 = sketch(capi, e0, reads_dev, 7)
        assert np.array_equal(ids, ids7) and np.array_equal(cnt, cnt7)            # batching does not matter
        assert ids.size == np.unique(ids).size and cnt.min() >= 1
        # the COMPILED REFERENCE's key count for this workload: oracle/_ref/metakssd dist -A -p 256 over the same 50 M reads as a
        # FASTQ file, whose (id, count) multiset the product's sketch equals (tools/check_fullsize_vs_ref.py,
        # profiles/r04_fullsize_vs_reference.json)
        assert ids.size == 1573525
        # expected accepted occurrences: 129 k-mers per read, 1/4096 of the inner substrings accepted
        total = int(cnt.astype(np.int64).sum())
        assert abs(total - N * 129 / 4096) < 5 * (N * 129 / 4096) ** 0.5

        # halves on two engines, merged: equals the single engine byte for byte
        half = N // 2
        for e in (e0, e1):
            e.begin(capi.MK_MODE_KOC)
        e0.push_reads_device(reads_dev, STRIDE, half, 0)
        e1.push_reads_device(reads_dev + half * STRIDE, STRIDE, N - half, half)
        d1 = e1.partial_count()
        bufs = []
        for nbytes in (8 * d1, 4 * d1, 8 * d1):
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
            bufs.append(p)
        assert e1.partial_export(bufs[0].value, bufs[1].value, bufs[2].value, d1) == d1
        counts1 = np.zeros(d1, np.uint32)
        assert hip.hipMemcpy(C.c_void_p(counts1.ctypes.data), bufs[1], C.c_size_t(4 * d1), 2) == 0
        d0 = e0.partial_count()
        e0.partial_import(bufs[0].value, bufs[1].value, bufs[2].value, d1)
        mids, mcnt = e0.finish()[0]
        for p in bufs:
            hip.hipFree(p)
        assert np.array_equal(mids, ids) and np.array_equal(mcnt, cnt)
        assert d0 + d1 >= ids.size and int(counts1.astype(np.int64).sum()) < total
    finally:
        e0.close()
        e1.close()


# ---- BASELINE config 4's table regime on ONE GPU -----------------------------------------------------------------------------
# 500 M synthetic 150 bp reads (80 GB of rows resident in HBM): 15 692 589 distinct keys in the 33 554 393-slot table (load 0.468,
# 78 % of hashlimit, /root/reference/iseq2comem.c:61,701-718 -- the heavy-collision regime where the slot layout really depends
# on first-occurrence order).  The sketch of the whole stream and the merge of 8 contiguous shards through
# mk_partial_export / mk_partial_import (what 8 GPUs do) must agree byte for byte.
N4, SHARDS4, DISTINCT4 = 500_000_000, 8, 15_692_589


def test_config4_regime_whole_equals_eight_shards(capi, shufs):
    import time
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    if free.value < N4 * STRIDE + (24 << 30):
        pytest.skip("needs %d GB of free HBM" % ((N4 * STRIDE + (24 << 30)) >> 30))
    shuf = shufs("L3K11")
    rows = C.c_void_p()
    assert hip.hipMalloc(C.byref(rows), C.c_size_t(N4 * STRIDE)) == 0
    e0, e1 = capi.Engine(shuf, 0), capi.Engine(shuf, 0)
    bufs = []
    try:
        capi.synth_rows_device(0, None, SEED, 0, N4, LEN, STRIDE, rows.value)
        assert hip.hipDeviceSynchronize() == 0
        per = 62_500_000  # rows per push: eight scan launches, as the bench does
        e0.begin(capi.MK_MODE_KOC)
        for first in range(0, N4, per):
            e0.push_reads_device(rows.value + first * STRIDE, STRIDE, min(per, N4 - first), first)
        e0.profile_enable(True)
        e0.profile_reset()
        t0 = time.time()
        ids, cnt = e0.finish()[0]
        t_finish = time.time() - t0
        prof = e0.profile()
        e0.profile_enable(False)
        msg = "finish %.2f ms wall, %.2f ms on the device, %d keys" % (1e3 * t_finish, prof["finish_ms"], ids.size)
        assert ids.size == DISTINCT4, msg
        assert abs(ids.size / 33554393.0 - 0.468) < 0.001, msg
        assert ids.size == np.unique(ids).size and cnt.min() >= 1, msg
        total_occ = int(cnt.astype(np.int64).sum())
        assert abs(total_occ - N4 * 129 / 4096) < 5 * (N4 * 129 / 4096) ** 0.5, msg

        # eight contiguous shards: shard 0 stays in e0's table, 1..7 are sketched on e1 one after the other, exported and
        # imported into e0 (what GPU 0 does with the lists of GPUs 1..7)
        shard = N4 // SHARDS4
        e0.begin(capi.MK_MODE_KOC)
        e0.push_reads_device(rows.value, STRIDE, shard, 0)
        cap = 4_000_000
        for nbytes in (8 * cap, 4 * cap, 8 * cap):
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
            bufs.append(p)
        partial_total = 0
        for s in range(1, SHARDS4):
            first = s * shard
            n = shard if s + 1 < SHARDS4 else N4 - first
            e1.begin(capi.MK_MODE_KOC)
            e1.push_reads_device(rows.value + first * STRIDE, STRIDE, n, first)
            d = e1.partial_count()
            assert d <= cap
            assert e1.partial_export(bufs[0].value, bufs[1].value, bufs[2].value, cap) == d
            partial_total += d
            e0.partial_import(bufs[0].value, bufs[1].value, bufs[2].value, d)
            e0.sync()  # the buffers are reused by the next shard
        t0 = time.time()
        mids, mcnt = e0.finish()[0]
        t_merge_finish = time.time() - t0
        msg += "; merged finish %.2f ms wall; partial lists of shards 1..7: %d keys" % (1e3 * t_merge_finish, partial_total)
        assert np.array_equal(mids, ids), msg
        assert np.array_equal(mcnt, cnt), msg
        assert partial_total + 1 >= ids.size * 7 // 8, msg
        print("\nconfig 4 regime on one GPU: " + msg)
    finally:
        for p in bufs:
            hip.hipFree(p)
        e0.close()
        e1.close()
        hip.hipFree(rows)
