#!/usr/bin/env python3
"""tools/search_multi.py -- `metakssd dist -r <mco dir> -o out [-M -O -N -D --correction] <sketch dir>` on N GPUs (SURVEY.md 8e
for the search, 8f N4): the query sketches are sharded across ranks in contiguous blocks, every rank holds the database's
genome lists and counts its block, rank 0 receives the rows in sketch order and writes sharedk_ct.dat and distance.out (the
same bytes the single-GPU CLI writes).  No collective on the data path, one gather of the result rows.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/search_multi.py -r refmco -o outdir [-M 0|1] [-O 0|1|2] [-N n] [-D d] [--correction 0|1] qrydir [--backend nccl|gloo] [--same-device]
"""
import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_stat(path, header):
    b = open(path, "rb").read()
    if header == 20:    # mco_dstat_t: shuf_id, kmerlen, dim_rd_len, comp_num, infile_num (command_dist.h:66-75)
        shuf_id, kmerlen, dim_rd_len, comp_num, n = struct.unpack_from("<Iiiii", b, 0)
    else:               # co_dstat_t (global_basic.h:116-126)
        shuf_id, = struct.unpack_from("<I", b, 0)
        kmerlen, dim_rd_len, comp_num, n = struct.unpack_from("<iiii", b, 8)
    cts = np.frombuffer(b, np.uint32, n, header)
    names = [b[header + 4 * n + 256 * i: header + 4 * n + 256 * (i + 1)].split(b"\0", 1)[0].decode() for i in range(n)]
    return dict(shuf_id=shuf_id, kmerlen=kmerlen, dim_rd_len=dim_rd_len, comp_num=comp_num, n=n), cts, names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-r", required=True)
    ap.add_argument("-o", default="./")
    ap.add_argument("-M", type=int, default=0)
    ap.add_argument("-O", type=int, default=2)
    ap.add_argument("-N", type=int, default=0)
    ap.add_argument("-D", type=float, default=1.0)
    ap.add_argument("--correction", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--same-device", action="store_true", help="debug: every rank on GPU 0 (gloo)")
    ap.add_argument("qry")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from metakssd_amd import capi
    from metakssd_amd.shard import gather_count_rows, shard_queries
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    rst, ref_ct, rnames = read_stat(os.path.join(a.r, "mcofiles.stat"), 20)
    qst, qry_ct, qnames = read_stat(os.path.join(a.qry, "cofiles.stat"), 32)
    if rst["shuf_id"] != qst["shuf_id"] or rst["comp_num"] != qst["comp_num"]:
        raise SystemExit("query sketch does not match the database (shuf_id / comp_num)")
    lo, hi = shard_queries(qst["n"], rank, world)
    m = capi.Mco(local)
    comps = []
    for c in range(rst["comp_num"]):
        gids = np.fromfile(os.path.join(a.r, "mco.%d" % c), np.uint32)
        index = np.memmap(os.path.join(a.r, "mco.index.%d" % c), np.uint64, "r")      # 2^32 cumulative row ends
        qindex = np.fromfile(os.path.join(a.qry, "combco.index.%d" % c), np.uint64)
        qids = np.fromfile(os.path.join(a.qry, "combco.%d" % c), np.uint32)[int(qindex[lo]):int(qindex[hi])]
        ee = np.asarray(index[qids]) if qids.size else np.zeros(0, np.uint64)          # command_dist.c:1040-1041
        prev = np.where(qids > 0, qids - 1, 0)
        es = np.where(qids > 0, np.asarray(index[prev]) if qids.size else 0, 0).astype(np.uint64)
        comps.append({"gids": gids, "ext_start": es, "ext_end": ee, "qry_index": (qindex[lo:hi + 1] - qindex[lo]).astype(np.uint64)})
    mine = m.count(rst["n"], None, qry_ct[lo:hi], comps) if hi > lo else np.zeros((0, rst["n"]), np.uint32)
    m.close()
    xdev = torch.device("cuda", local) if (world > 1 and a.backend == "nccl") else torch.device("cpu")
    ct = gather_count_rows(mine, qst["n"], rst["n"], dst=0, device=xdev) if world > 1 else mine
    if rank == 0:
        os.makedirs(a.o, exist_ok=True)
        ct.astype(np.uint32).tofile(os.path.join(a.o, "sharedk_ct.dat"))
        rc = capi.dist_print(os.path.join(a.o, "distance.out"), ref_ct, qry_ct, rnames, qnames, ct, qst["kmerlen"], qst["dim_rd_len"],
                             metric=a.M, outfields=a.O, correction=a.correction, num_neigb=a.N, dthreshold=a.D)
        if rc:
            raise SystemExit("mk_dist_print failed (%d)" % rc)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
