"""Read-range sharding and the one exchange step of the multi-GPU sketch.

Reads shard as contiguous ranges with GLOBAL ordinals; each rank keeps a private table, so there is no
data-path collective during the scan.  The only exchange is a gather of the per-rank distinct-key lists
{key u64, count u32 (clamped to 65535), first ordinal u64} to rank 0, which folds them into its table
(counts add, first ordinals take min) and rebuilds the sequential layout -- or, from four ranks on, SURVEY.md 8e's alternative:
an all-to-all by key % world (exchange_slices), every rank folds a key slice, and the gather moves REDUCED slices, which are
rank 0's key list as they stand.  With the "nccl" backend this is
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


def gather_partials_concat(keys, counts, ords, n, dst=0, group=None, out=None):
    """the same exchange with the received lists landing back to back in out = (keys, counts, ords) on `dst` (tensors of
    the exchange device with room for every rank's list), so that one mk_partial_import launch folds them all in.
    Returns the number of entries received on `dst`, 0 elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = keys.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    ops, total = [], 0
    if rank == dst:
        rk, rc, ro = out
        incoming = sum(sizes[r] for r in range(world) if r != dst)
        if incoming > rk.numel():
            raise ValueError("gather_partials_concat: %d incoming entries, room for %d" % (incoming, rk.numel()))
        for r in range(world):
            if r == dst or sizes[r] == 0:
                continue
            a, b = total, total + sizes[r]
            ops += [dist.P2POp(dist.irecv, rk[a:b], r, group), dist.P2POp(dist.irecv, rc[a:b], r, group),
                    dist.P2POp(dist.irecv, ro[a:b], r, group)]
            total = b
    elif n > 0:
        ops += [dist.P2POp(dist.isend, keys[:n].contiguous(), dst, group),
                dist.P2POp(dist.isend, counts[:n].contiguous(), dst, group),
                dist.P2POp(dist.isend, ords[:n].contiguous(), dst, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return total


# ---- SURVEY.md 8e's alternative exchange: all-to-all by key % world, every rank reduces a key slice, gather of the reduced slices ----
def exchange_slices(keys, counts, ords, part_sizes, group=None, out=None):
    """This rank's distinct list arrives CUT INTO world PARTS by key % world (mk_partial_export_split: part g starts at
    sum(part_sizes[:g])).  Part g of every rank goes to rank g.  Returns (n_received, (keys, counts, ords)): the parts this rank
    received from the OTHER ranks, back to back in rank order, in `out` when given (tensors of the exchange device with room for
    them) or in fresh tensors.  The rank's own part stays where it is (the caller folds it in from there).  One exchange: an
    all-gather of the world x world part sizes, then batched isend / irecv (RCCL point-to-point with the "nccl" backend)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = keys.device
    assert len(part_sizes) == world
    mine = torch.tensor([int(x) for x in part_sizes], dtype=torch.int64, device=dev)
    allp = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allp, mine, group=group)
    sizes = [[int(v) for v in t.tolist()] for t in allp]  # sizes[r][g]: what rank r holds for rank g
    incoming = sum(sizes[r][rank] for r in range(world) if r != rank)
    if out is None:
        out = (torch.empty(incoming, dtype=torch.int64, device=dev), torch.empty(incoming, dtype=torch.int32, device=dev),
               torch.empty(incoming, dtype=torch.int64, device=dev))
    rk, rc, ro = out
    if incoming > rk.numel():
        raise ValueError("exchange_slices: %d incoming entries, room for %d" % (incoming, rk.numel()))
    ops, at = [], 0
    for r in range(world):  # receives in rank order
        n = sizes[r][rank]
        if r == rank or n == 0:
            continue
        ops += [dist.P2POp(dist.irecv, rk[at:at + n], r, group), dist.P2POp(dist.irecv, rc[at:at + n], r, group),
                dist.P2POp(dist.irecv, ro[at:at + n], r, group)]
        at += n
    off = 0
    for g in range(world):
        n = sizes[rank][g]
        if g != rank and n:
            ops += [dist.P2POp(dist.isend, keys[off:off + n].contiguous(), g, group),
                    dist.P2POp(dist.isend, counts[off:off + n].contiguous(), g, group),
                    dist.P2POp(dist.isend, ords[off:off + n].contiguous(), g, group)]
        off += n
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return incoming, out


# ---- config 5: whole input files are the unit -- no reduction, only a gather in file order -----------
def shard_files(nfiles, rank, world):
    """indices of the files `rank` sketches: contiguous blocks, so that rank order == file order"""
    lo, hi = shard_range(nfiles, rank, world)
    return list(range(lo, hi))


def gather_file_sketches(mine, ncomp, dst=0, group=None, device=None):
    """mine: list of per-file sketches of this rank, each a list over components of uint32 numpy arrays (ids).
    Returns on `dst` the list of ALL files' sketches in rank order (= file order with shard_files); [] elsewhere.
    One exchange: all_gather of the per-file, per-component lengths, then one id buffer per sending rank."""
    import numpy as np
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device or torch.device("cpu")
    lens = np.array([[len(c) for c in f] for f in mine], dtype=np.int64).reshape(-1, ncomp)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([lens.shape[0]], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    maxf = max(counts) if counts else 0
    pad = torch.zeros(max(1, maxf) * ncomp, dtype=torch.int64, device=dev)
    pad[: lens.size] = torch.from_numpy(lens.reshape(-1)).to(dev)
    all_lens = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(all_lens, pad, group=group)
    flat = np.concatenate([c for f in mine for c in f]).astype(np.uint32) if lens.size and lens.sum() else np.zeros(0, np.uint32)
    ops, bufs = [], {}
    if rank == dst:
        for r in range(world):
            n = int(all_lens[r][: counts[r] * ncomp].sum().item())
            if r != dst and n:
                bufs[r] = torch.empty(n, dtype=torch.int32, device=dev)
                ops.append(dist.P2POp(dist.irecv, bufs[r], r, group))
    elif flat.size:
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(flat.view(np.int32).copy()).to(dev), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return []
    out = []
    for r in range(world):
        data = flat if r == dst else (bufs[r].cpu().numpy().view(np.uint32) if r in bufs else np.zeros(0, np.uint32))
        ll = all_lens[r][: counts[r] * ncomp].cpu().numpy().reshape(-1, ncomp)
        at = 0
        for f in range(counts[r]):
            comps = []
            for c in range(ncomp):
                comps.append(data[at: at + int(ll[f, c])].copy())
                at += int(ll[f, c])
            out.append(comps)
    return out


# ---- `dist -r` search: query sketches are the unit -- every rank counts a contiguous block of query
# sketches against the whole database; the rows of the count matrix are gathered in rank order, nothing is reduced -----
def shard_queries(qry_num, rank, world):
    """[lo, hi) of the query sketches `rank` counts (contiguous, rank order == sketch order)"""
    return shard_range(qry_num, rank, world)


def gather_count_rows(local_ct, qry_num, ref_num, dst=0, group=None, device=None):
    """local_ct: uint32 numpy array (my query sketches x ref_num).  Returns on `dst` the whole qry_num x ref_num matrix
    (rows in sketch order), None elsewhere.  One exchange: every other rank sends its block to `dst`."""
    import numpy as np
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device or torch.device("cpu")
    local_ct = np.ascontiguousarray(local_ct, dtype=np.uint32).reshape(-1, ref_num) if ref_num else np.zeros((0, 0), np.uint32)
    lo, hi = shard_queries(qry_num, rank, world)
    assert local_ct.shape[0] == hi - lo or ref_num == 0
    ops, bufs = [], {}
    if rank == dst:
        for r in range(world):
            rlo, rhi = shard_queries(qry_num, r, world)
            n = (rhi - rlo) * ref_num
            if r != dst and n:
                bufs[r] = torch.empty(n, dtype=torch.int32, device=dev)
                ops.append(dist.P2POp(dist.irecv, bufs[r], r, group))
    elif local_ct.size:
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(local_ct.reshape(-1).view(np.int32).copy()).to(dev), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return None
    out = np.zeros((qry_num, ref_num), np.uint32)
    for r in range(world):
        rlo, rhi = shard_queries(qry_num, r, world)
        if rhi > rlo and ref_num:
            out[rlo:rhi] = local_ct if r == dst else bufs[r].cpu().numpy().view(np.uint32).reshape(rhi - rlo, ref_num)
    return out
