"""worker for tests/test_dist_gloo.py: one rank of the sharded sketch over gloo (CPU tensors).

Each rank sketches its contiguous read range with GLOBAL ordinals (here with the oracle's shard model, because the
product's scan needs a GPU), the product's gather_partials() moves the lists to rank 0, rank 0 merges and lays out.
Checked: (1) shard ranges tile the input, (2) the exchange delivers every list intact, (3) merged == sequential;
(4) the same through SURVEY.md 8e's alternative exchange (exchange_slices: all-to-all by key % world, every rank folds a key
slice, gather of the reduced slices): every rank receives exactly its slice, the concatenated slices == sequential."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metakssd_amd import capi
    from metakssd_amd.shard import gather_partials, gather_partials_concat, shard_range
    from oracle_binding import Oracle
    import util_inputs as ui

    shuf = capi.Shuf.generate(6, 3, 0, 6)  # accept-everything table: the merge has real collisions to resolve
    rs = np.random.RandomState(17)
    seqs = ui.pool_reads(rs, 6000, 1201)   # odd count: uneven shards
    rows = ui.rows_from_seqs(seqs, 160)
    total = len(seqs)
    lo, hi = shard_range(total, rank, world)
    ranges = [shard_range(total, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == total and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))

    ora = Oracle(shuf.c.id, 6, 3, 0, shuf.table)
    keys, cnts, ords = ora.partial_from_rows(rows[lo * 160:hi * 160], 160, lo)
    tk = torch.from_numpy(keys.view(np.int64).copy())
    tc = torch.from_numpy(cnts.view(np.int32).copy())
    to = torch.from_numpy(ords.view(np.int64).copy())
    got = gather_partials(tk, tc, to, len(keys), dst=0)
    ok = True
    # the same exchange with everything landing back to back in preallocated buffers (what bench.py's rank 0 imports with
    # one launch): same entries in rank order
    cap = 200000
    outbuf = (torch.empty(cap, dtype=torch.int64), torch.empty(cap, dtype=torch.int32), torch.empty(cap, dtype=torch.int64))
    if rank == 0:
        tot = gather_partials_concat(tk[:0], tc[:0], to[:0], 0, dst=0, out=outbuf)
        cat_k = torch.cat([k for (k, c, o) in got]) if got else tk[:0]
        cat_c = torch.cat([c for (k, c, o) in got]) if got else tc[:0]
        cat_o = torch.cat([o for (k, c, o) in got]) if got else to[:0]
        ok &= tot == cat_k.numel() and torch.equal(outbuf[0][:tot], cat_k) and torch.equal(outbuf[1][:tot], cat_c) and torch.equal(outbuf[2][:tot], cat_o)
    else:
        assert gather_partials_concat(tk, tc, to, len(keys), dst=0) == 0
    if rank == 0:
        parts = [(keys, cnts, ords)]
        for (k, c, o) in got:
            parts.append((k.numpy().view(np.uint64), c.numpy().view(np.uint32), o.numpy().view(np.uint64)))
        assert len(parts) == world
        # every received list must equal what that rank computed (recompute here)
        for r in range(1, world):
            rlo, rhi = ranges[r]
            ek, ec, eo = Oracle(shuf.c.id, 6, 3, 0, shuf.table).partial_from_rows(rows[rlo * 160:rhi * 160], 160, rlo)
            pk, pc, po = parts[r]
            ok &= np.array_equal(pk, ek) and np.array_equal(pc, ec) and np.array_equal(po, eo)
        merged = ora.layout_from_partials(parts)
        rc, want = Oracle(shuf.c.id, 6, 3, 0, shuf.table).koc_from_rows(rows, 160)
        ok &= rc == 0 and np.array_equal(merged[0][0], want[0][0]) and np.array_equal(merged[0][1], want[0][1])
        ok &= len(want[0][0]) > 1000
        open(os.environ["MK_DIST_RESULT"], "w").write("OK %d" % len(want[0][0]) if ok else "MISMATCH")
    else:
        assert got == []

    # ---- SURVEY.md 8e's alternative: all-to-all by key % world, every rank reduces a key slice, gather of the reduced slices ----
    from metakssd_amd.shard import exchange_slices
    order = np.argsort(keys % np.uint64(world), kind="stable")  # the list cut into parts by key % world (mk_partial_export_split)
    sk, sc, so = keys[order], cnts[order], ords[order]
    part_sizes = [int(np.sum(keys % np.uint64(world) == g)) for g in range(world)]
    n_in, (rk, rc_, ro) = exchange_slices(torch.from_numpy(sk.view(np.int64).copy()), torch.from_numpy(sc.view(np.int32).copy()),
                                          torch.from_numpy(so.view(np.int64).copy()), part_sizes)
    own0 = sum(part_sizes[:rank])
    ak = np.concatenate([sk[own0:own0 + part_sizes[rank]], rk[:n_in].numpy().view(np.uint64)])
    ac = np.concatenate([sc[own0:own0 + part_sizes[rank]], rc_[:n_in].numpy().view(np.uint32)])
    ao = np.concatenate([so[own0:own0 + part_sizes[rank]], ro[:n_in].numpy().view(np.uint64)])
    assert np.all(ak % np.uint64(world) == rank)  # this rank's slice, and nothing else
    # fold the slice: counts add (clamped), first ordinals take the minimum -- what mk_partial_import does in the table
    uk, inv = np.unique(ak, return_inverse=True)
    uc = np.minimum(np.bincount(inv, weights=ac.astype(np.float64), minlength=len(uk)), 65535).astype(np.uint32)
    uo = np.full(len(uk), np.iinfo(np.uint64).max, np.uint64)
    np.minimum.at(uo, inv, ao)
    tk2, tc2, to2 = torch.from_numpy(uk.view(np.int64).copy()), torch.from_numpy(uc.view(np.int32).copy()), torch.from_numpy(uo.view(np.int64).copy())
    if rank == 0:
        tot = gather_partials_concat(tk2[:0], tc2[:0], to2[:0], 0, dst=0, out=outbuf)
        # the slices are disjoint key sets: their concatenation is the sketch's list of distinct keys, nothing left to fold
        lk = np.concatenate([uk, outbuf[0][:tot].numpy().view(np.uint64)])
        lc = np.concatenate([uc, outbuf[1][:tot].numpy().view(np.uint32)])
        lo_ = np.concatenate([uo, outbuf[2][:tot].numpy().view(np.uint64)])
        ok2 = len(np.unique(lk)) == len(lk)
        merged2 = Oracle(shuf.c.id, 6, 3, 0, shuf.table).layout_from_partials([(lk, lc, lo_)])
        ok2 &= np.array_equal(merged2[0][0], want[0][0]) and np.array_equal(merged2[0][1], want[0][1])
        if not ok2:
            open(os.environ["MK_DIST_RESULT"], "w").write("MISMATCH (slices)")
    else:
        assert gather_partials_concat(tk2, tc2, to2, len(uk), dst=0) == 0
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
