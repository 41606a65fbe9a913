#!/usr/bin/env python3
"""tools/dist_genomes_multi.py -- BASELINE config 5 on N GPUs: `metakssd dist -L x.shuf [-u] -o out <genome dir>`
with the genome files sharded across ranks in contiguous blocks.  Whole files are the unit: no reduction, one gather of the
per-file id arrays to rank 0, which writes the sketch directory in file order (the same bytes the single-GPU CLI writes).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/dist_genomes_multi.py -L L3K10.shuf [-u] -o outdir genomes_dir [--backend nccl|gloo] [--same-device]
"""
import argparse
import ctypes as C
import gzip
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FASTA_EXT = (".fasta", ".fa", ".fna", ".fas")


def is_fasta(name):
    n = name[:-3] if name.endswith(".gz") else name
    return n.lower().endswith(FASTA_EXT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-L", required=True)
    ap.add_argument("-o", default="./")
    ap.add_argument("-u", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--same-device", action="store_true", help="debug: every rank on GPU 0 (gloo)")
    ap.add_argument("dir")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from metakssd_amd import capi
    from metakssd_amd.shard import gather_file_sketches, shard_files
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    files = sorted(os.path.join(a.dir, f) for f in os.listdir(a.dir) if is_fasta(f))  # the CLI's discovery order
    shuf = capi.Shuf.read(a.L)
    P = shuf.params()
    eng = capi.Engine(shuf, local)
    mode = capi.MK_MODE_UNIQ_SET if a.u else capi.MK_MODE_SET
    t0 = time.perf_counter()
    mine = []
    for i in (shard_files(len(files), rank, world) if world > 1 else range(len(files))):
        raw = open(files[i], "rb").read()
        data = gzip.decompress(raw) if files[i].endswith(".gz") else raw
        rows = capi.fasta_windows(data, P.TL, 512)
        eng.begin(mode)
        eng.push_reads(rows, 512, 0)
        mine.append([ids for ids, _ in eng.finish()])
    xdev = torch.device("cuda", local) if (world > 1 and a.backend == "nccl") else torch.device("cpu")
    allf = gather_file_sketches(mine, P.component_num, dst=0, device=xdev) if world > 1 else mine
    if rank == 0:
        assert len(allf) == len(files)
        sd = C.c_void_p()
        pc = capi.ParamsC()
        C.memmove(C.byref(pc), C.byref(P), C.sizeof(pc))
        rc = capi.lib.mk_sketchdir_open(a.o.encode(), C.byref(pc), 0, len(files), C.byref(sd))
        assert rc == 0, rc
        for path, comps in zip(files, allf):
            arr = (capi.ComponentC * len(comps))()
            keep = []
            for c, ids in enumerate(comps):
                ids = np.ascontiguousarray(ids, np.uint32)
                keep.append(ids)
                arr[c].ids = ids.ctypes.data_as(C.POINTER(C.c_uint32)) if ids.size else None
                arr[c].counts = None
                arr[c].n = ids.size
            res = capi.ResultC(len(comps), sum(int(x.size) for x in keep), arr)
            assert capi.lib.mk_sketchdir_add(sd, path.encode(), C.byref(res)) == 0
        assert capi.lib.mk_sketchdir_close(sd) == 0
        dt = time.perf_counter() - t0
        print("sketched %d genomes on %d GPU(s) in %.2f s = %.1f genomes/s" % (len(files), world, dt, len(files) / dt))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
