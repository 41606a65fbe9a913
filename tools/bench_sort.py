#!/usr/bin/env python3
"""Stage II's device sort alone: n random (id, genome) pairs through mk_mco_sort_pairs, checked against numpy's stable sort.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel times (mk_rs_*); MK_LIBRARY selects another build."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metakssd_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=100_000_000)
ap.add_argument("--key-bits", type=int, default=32)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-verify", action="store_true", help="skip the numpy check (minutes at 400 M pairs); sortedness is still checked")
a = ap.parse_args()
rng = np.random.default_rng(5)
k = rng.integers(0, 1 << a.key_bits, a.pairs, dtype=np.uint64).astype(np.uint32)
v = np.arange(a.pairs, dtype=np.uint32)
m = capi.Mco(0)
for r in range(a.reps):
    t0 = time.time()
    sk, sv = m.sort_pairs(k, v)
    print("rep %d: %.1f ms (H2D + sort + D2H)" % (r, 1e3 * (time.time() - t0)), flush=True)
if a.no_verify:
    assert np.all(sk[1:] >= sk[:-1]) and np.array_equal(k[sv[::997]], sk[::997]), "not sorted / values detached from keys"
    print("ok (sortedness + sampled key/value pairing): %d pairs" % a.pairs)
    sys.exit(0)
order = np.argsort(k, kind="stable")
assert np.array_equal(sk, k[order]) and np.array_equal(sv, v[order]), "sort differs from numpy's stable sort"
print("ok: %d pairs, %d key bits" % (a.pairs, a.key_bits))
