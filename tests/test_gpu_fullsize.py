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
        assert ids.size == 1573525                                                # bench.py's distinct_keys for this workload
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
