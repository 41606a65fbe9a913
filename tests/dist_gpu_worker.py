"""worker for tests/test_gpu_parity.py::test_engine_export_import_across_processes: one rank of the sharded sketch with the PRODUCT's
engine on the GPU (every rank on GPU 0: this is a one-GPU box) and the lists moved by gloo through host memory.

What tests/dist_worker.py cannot show (its lists are oracle-made): mk_partial_export / _export_split / _restart / _import /
_list_adopt / _list_commit of engines living in SEVERAL PROCESSES, fed by the real scan, merged on rank 0 -- checked there against the
oracle's sequential sketch of all reads (ids and counts, in order).  MK_DIST_MERGE = gather | slices."""
import ctypes as C
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
    merge = os.environ.get("MK_DIST_MERGE", "gather")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metakssd_amd import capi
    from metakssd_amd.shard import exchange_slices, gather_partials_concat, shard_range
    import util_inputs as ui

    hip = C.CDLL("libamdhip64.so")
    shuf = capi.Shuf.generate(7, 4, 1, 7)
    rs = np.random.RandomState(23)
    seqs = ui.pool_reads(rs, 30000, 9001) + ui.ragged_reads(rs, 150)
    stride = 304
    rows = ui.rows_from_seqs(seqs, stride)
    total = len(seqs)
    lo, hi = shard_range(total, rank, world)
    eng = capi.Engine(shuf, 0)
    cap = int(eng.params.hashlimit) + 1

    def dbuf(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        return p.value

    def to_host(ptr, n, dtype):
        a = np.zeros(n, dtype)
        if n:
            assert hip.hipMemcpy(C.c_void_p(a.ctypes.data), C.c_void_p(ptr), C.c_size_t(a.nbytes), 2) == 0
        return a

    def to_dev(ptr, a):
        if a.size:
            assert hip.hipMemcpy(C.c_void_p(ptr), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0

    pk, pc, po = dbuf(8 * cap), dbuf(4 * cap), dbuf(8 * cap)   # this rank's export
    rk, rc, ro = dbuf(8 * cap), dbuf(4 * cap), dbuf(8 * cap)   # what it receives
    eng.begin(capi.MK_MODE_KOC)
    eng.push_reads(rows[lo * stride:hi * stride], stride, lo)

    def tensors(n, k=pk, c=pc, o=po):
        return (torch.from_numpy(to_host(k, n, np.uint64).view(np.int64)), torch.from_numpy(to_host(c, n, np.uint32).view(np.int32)),
                torch.from_numpy(to_host(o, n, np.uint64).view(np.int64)))
    big = 400000
    outbuf = (torch.empty(big, dtype=torch.int64), torch.empty(big, dtype=torch.int32), torch.empty(big, dtype=torch.int64))
    merged = None
    if merge == "gather":
        if rank != 0:
            m = eng.partial_export(pk, pc, po, cap)
            assert gather_partials_concat(*tensors(m), m, dst=0) == 0
        else:
            tot = gather_partials_concat(outbuf[0][:0], outbuf[1][:0], outbuf[2][:0], 0, dst=0, out=outbuf)
            to_dev(rk, outbuf[0][:tot].numpy()); to_dev(rc, outbuf[1][:tot].numpy()); to_dev(ro, outbuf[2][:tot].numpy())
            eng.partial_import(rk, rc, ro, tot)
            merged = eng.finish()
    else:
        d, parts = eng.partial_export_split(world, pk, pc, po, cap)
        eng.partial_restart()
        n_in, (sk, sc, so) = exchange_slices(*tensors(d), parts)
        own_at, own = sum(parts[:rank]), parts[rank]
        if own:
            eng.partial_import(pk + 8 * own_at, pc + 4 * own_at, po + 8 * own_at, own)
        to_dev(rk, sk[:n_in].numpy()); to_dev(rc, sc[:n_in].numpy()); to_dev(ro, so[:n_in].numpy())
        if n_in:
            eng.partial_import(rk, rc, ro, n_in)
        if rank != 0:
            m = eng.partial_export(pk, pc, po, cap)  # the reduced slice
            assert gather_partials_concat(*tensors(m), m, dst=0) == 0
        else:
            r0 = eng.partial_count()
            tot = gather_partials_concat(outbuf[0][:0], outbuf[1][:0], outbuf[2][:0], 0, dst=0, out=outbuf)
            to_dev(rk, outbuf[0][:tot].numpy()); to_dev(rc, outbuf[1][:tot].numpy()); to_dev(ro, outbuf[2][:tot].numpy())
            eng.partial_list_adopt(rk, rc, ro, tot, r0)
            eng.partial_list_commit(r0 + tot)
            merged = eng.finish()
    if rank == 0:
        from oracle_binding import Oracle
        rcode, want = Oracle(shuf.c.id, 7, 4, 1, shuf.table).koc_from_rows(rows, stride)
        ok = rcode == 0 and len(merged) == len(want) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(merged, want))
        ok = ok and len(want[0][0]) > 1000
        open(os.environ["MK_DIST_RESULT"], "w").write("OK %d" % len(want[0][0]) if ok else "MISMATCH")
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
