"""Read-range sharding and the one exchange step of the multi-GPU sketch.

Reads shard as contiguous ranges with GLOBAL ordinals; each rank keeps a private table, so there is no
data-path collective during the scan.  The only exchange is a gather of the per-rank distinct-key lists
{key u64, count u32 (clamped to 65535), first ordinal u64} to rank 0, which folds them into its table
(counts add, first ordinals take min) and rebuilds the sequential layout.  With the "nccl" backend this is
RCCL point-to-point over xGMI: every sender has its own link to rank 0, lists are a few MB to tens of MB.
The same code runs over gloo with CPU tensors (tests).
"""
import torch
import torch.distributed as dist


def shard_range(total_reads, rank, world):
    """contiguous range [lo, hi) of reads for `rank`; ranges are ordered by rank so ordinals stay global"""
    base, rem = divmod(total_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_partials(keys, counts, ords, n, dst=0, group=None):
    """keys/ords: int64 tensors (bit patterns of the u64 values), counts: int32 tensor, first n entries valid.
    Returns on `dst` a list over ranks != dst of (keys, counts, ords) tensors; on other ranks []."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = keys.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    ops, out = [], []
    if rank == dst:
        for r in range(world):
            if r == dst or sizes[r] == 0:
                continue
            k = torch.empty(sizes[r], dtype=torch.int64, device=dev)
            c = torch.empty(sizes[r], dtype=torch.int32, device=dev)
            o = torch.empty(sizes[r], dtype=torch.int64, device=dev)
            out.append((k, c, o))
            ops += [dist.P2POp(dist.irecv, k, r, group), dist.P2POp(dist.irecv, c, r, group),
                    dist.P2POp(dist.irecv, o, r, group)]
    elif n > 0:
        ops += [dist.P2POp(dist.isend, keys[:n].contiguous(), dst, group),
                dist.P2POp(dist.isend, counts[:n].contiguous(), dst, group),
                dist.P2POp(dist.isend, ords[:n].contiguous(), dst, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out
