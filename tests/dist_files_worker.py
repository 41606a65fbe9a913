"""worker for tests/test_dist_gloo.py: config 5's multi-GPU shape over gloo -- files sharded in contiguous blocks, per-file
sketches (here: seeded fake id arrays, the sketching itself needs a GPU) gathered to rank 0 in file order."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fake_sketch(i, ncomp):
    rs = np.random.RandomState(1000 + i)
    return [rs.randint(0, 2 ** 32, size=(0 if (i + c) % 5 == 0 else 10 + 7 * i + c), dtype=np.uint64).astype(np.uint32) for c in range(ncomp)]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metakssd_amd.shard import gather_file_sketches, shard_files
    ok = True
    for nfiles, ncomp in ((11, 1), (5, 16), (1, 2), (0, 1)):
        idx = shard_files(nfiles, rank, world)
        mine = [fake_sketch(i, ncomp) for i in idx]
        got = gather_file_sketches(mine, ncomp, dst=0)
        if rank == 0:
            ok &= len(got) == nfiles
            for i in range(min(len(got), nfiles)):
                want = fake_sketch(i, ncomp)
                ok &= len(got[i]) == ncomp and all(np.array_equal(a, b) for a, b in zip(got[i], want))
        else:
            ok &= got == []
    if rank == 0:
        open(os.environ["MK_DIST_RESULT"], "w").write("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
