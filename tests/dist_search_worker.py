"""worker for tests/test_dist_gloo.py: the search's multi-GPU shape over gloo -- query sketches sharded in contiguous blocks,
every rank counts its block against the whole database (here with the oracle's counting loop: the device kernel needs a
GPU), the rows of the matrix are gathered on rank 0 and must equal the unsharded matrix."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    from metakssd_amd.shard import gather_count_rows, shard_queries
    ok = True
    for nref, nqry in ((5, 7), (3, 1), (4, 0), (9, 2), (1, 12)):
        rs = np.random.RandomState(nref * 31 + nqry)      # the same data on every rank
        pool = np.unique(rs.randint(0, 2 ** 32, size=3000, dtype=np.uint64).astype(np.uint32))

        def draw(n):
            parts, index = [], [0]
            for _ in range(n):
                p = rs.permutation(pool)[:int(rs.randint(0, 800))]
                parts.append(p)
                index.append(index[-1] + p.size)
            return (np.concatenate(parts) if parts else np.zeros(0, np.uint32)).astype(np.uint32), np.array(index, np.uint64)
        rids, rindex = draw(nref)
        qids, qindex = draw(nqry)
        ctx = np.diff(qindex).astype(np.uint32)
        gids, ri, re_ = ob.mco_build(rids, rindex)
        whole = ob.mco_count(gids, ri, re_, qids, qindex, ctx, nref)
        lo, hi = shard_queries(nqry, rank, world)
        sub_index = (qindex[lo:hi + 1] - qindex[lo]).astype(np.uint64)
        mine = ob.mco_count(gids, ri, re_, qids[int(qindex[lo]):int(qindex[hi])], sub_index, ctx[lo:hi], nref)
        got = gather_count_rows(mine, nqry, nref, dst=0)
        if rank == 0:
            ok &= got is not None and got.shape == whole.shape and np.array_equal(got, whole)
        else:
            ok &= got is None
    if rank == 0:
        open(os.environ["MK_DIST_RESULT"], "w").write("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
