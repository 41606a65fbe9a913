#!/usr/bin/env python3
"""bench.py -- Gbases/s sketched on synthetic 150 bp reads with an L3K11 .shuf (BASELINE.json metric).

One step = one whole sketch of the rank's resident read shard: table clear (mk_sketch_begin), scan of
every read (mk_sketch_push_reads_device, reads already in HBM), and finish (distinct-key compaction,
reference-order layout, slot-order dump, result copied to the host).  With N > 1 ranks (one process per
GPU, launched by torch.distributed.run) each rank scans its own contiguous read range with global
ordinals, ranks != 0 send their distinct-key lists to rank 0 over RCCL, rank 0 folds them in and
finishes -- "weak" scaling: per-GPU reads fixed.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the scan kernel
(HIP-event timed inside the engine on its launch stream) and, at N=1, `cpu_baseline`: the compiled
reference (oracle/_ref/metakssd, kind "reference") or the oracle port timed on this host's cores over a
bounded sample of the same workload.
"""
import argparse
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 150
STRIDE = 160
SEED = 20261002
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def cpu_baseline(shuf, sample_reads, gpu_sketch):
    """time the reference's OpenMP CPU path on this host over `sample_reads` reads of the same workload"""
    import numpy as np
    from metakssd_amd import capi
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "metakssd")
    tmp = tempfile.mkdtemp(prefix="mkbench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        if os.path.exists(ref):
            fq = os.path.join(tmp, "sample.fq")
            sp = os.path.join(tmp, "L3K11.shuf")
            shuf.write(sp)
            rc = capi.lib.mk_synth_fastq_write(fq.encode(), SEED, 0, sample_reads, READ_LEN)
            assert rc == 0
            out = os.path.join(tmp, "out")
            t0 = time.perf_counter()
            r = subprocess.run([ref, "dist", "-L", sp, "-A", "-p", str(cores), "-o", out, fq],
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            if r.returncode != 0 or not os.path.exists(os.path.join(out, "cofiles.stat")):
                raise RuntimeError("reference run failed: " + r.stderr.decode(errors="replace")[-300:])
            ids = np.fromfile(os.path.join(out, "combco.0"), dtype=np.uint32)
            cnt = np.fromfile(os.path.join(out, "combco.0.a"), dtype=np.uint16)
            # -p N output order is not reproducible: compare as sorted (id,count) multisets
            a = np.sort(ids.astype(np.uint64) << np.uint64(16) | cnt.astype(np.uint64))
            b = np.sort(gpu_sketch[0][0].astype(np.uint64) << np.uint64(16) | gpu_sketch[0][1].astype(np.uint64))
            return {"value": sample_reads * READ_LEN / dt / 1e9, "unit": "Gbases/s", "cores": cores, "kind": "reference",
                    "sample": "first %d reads of the workload as FASTQ on tmpfs, `metakssd dist -L L3K11.shuf -A -p %d` "
                              "(compiled reference, wall %.2f s)" % (sample_reads, cores, dt),
                    "gpu_equals_reference_multiset": bool(np.array_equal(a, b))}
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_binding import Oracle
        ora = Oracle(shuf.c.id, shuf.c.k, shuf.c.subk, shuf.c.drlevel, shuf.table)
        rows = capi.synth_rows_host(SEED, 0, sample_reads, READ_LEN, STRIDE)
        t0 = time.perf_counter()
        ora.koc_from_rows_omp(rows, STRIDE, cores)
        dt = time.perf_counter() - t0
        return {"value": sample_reads * READ_LEN / dt / 1e9, "unit": "Gbases/s", "cores": cores, "kind": "port",
                "sample": "first %d reads of the workload, oracle OpenMP port of mt_shortreads2koc on pre-framed rows "
                          "(no FASTQ parsing), wall %.2f s" % (sample_reads, dt)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-gpu", type=int, default=50_000_000, help="BASELINE config 3: 50 M reads on one GPU")
    ap.add_argument("--cpu-sample-reads", type=int, default=4_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured configuration); gloo moves the lists through the host (debug)")
    ap.add_argument("--same-device", action="store_true", help="debug: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--verify", action="store_true",
                    help="after timing, rank 0 re-sketches ALL ranks' reads on one engine and compares with the merged result")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from metakssd_amd import capi
    from metakssd_amd.shard import gather_partials

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" %
                         (args.gpus, world, args.gpus))
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the exchanged lists live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    shuf = capi.Shuf.generate(11, 6, 3, 11)  # L3K11 = {k=11, subk=6, drlevel=3}, same bytes as the tests' table
    eng = capi.Engine(shuf, local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    eng.set_stream(stream)

    n = args.reads_per_gpu
    first = rank * n  # contiguous global read ranges, rank-ordered
    reads = torch.empty(n * STRIDE, dtype=torch.uint8, device=dev)
    capi.synth_rows_device(local_rank, stream, SEED, first, n, READ_LEN, STRIDE, reads.data_ptr())
    torch.cuda.synchronize()

    cap = eng.params.hashlimit + 1
    if world > 1:
        pk = torch.empty(cap, dtype=torch.int64, device=dev)
        pc = torch.empty(cap, dtype=torch.int32, device=dev)
        po = torch.empty(cap, dtype=torch.int64, device=dev)

    result = {}
    flags = {"keep": False}

    def step():
        eng.begin(capi.MK_MODE_KOC)
        eng.push_reads_device(reads.data_ptr(), STRIDE, n, first)
        if world > 1:
            if rank != 0:
                m = eng.partial_export(pk.data_ptr(), pc.data_ptr(), po.data_ptr(), cap)
                gather_partials(pk[:m].to(xdev), pc[:m].to(xdev), po[:m].to(xdev), m, dst=0)
            else:
                empty = (pk[:0].to(xdev), pc[:0].to(xdev), po[:0].to(xdev))
                for (k, c, o) in gather_partials(*empty, 0, dst=0):
                    k, c, o = k.to(dev), c.to(dev), o.to(dev)
                    # the engine runs on torch's current stream: the import is ordered after the receives/copies;
                    # the tensors must outlive the import kernel, hence the engine-side sync before they are dropped
                    torch.cuda.current_stream().synchronize()
                    eng.partial_import(k.data_ptr(), c.data_ptr(), o.data_ptr(), k.numel())
                    eng.sync()
        if rank == 0:
            if flags["keep"]:
                result["sketch"] = eng.finish()
                result["distinct"] = sum(len(x[0]) for x in result["sketch"])
            else:
                r = eng.finish_raw()
                result["distinct"] = int(r.total)
                capi.lib.mk_result_release(eng.h, r)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.profile_enable(True)
    eng.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = eng.profile()
    eng.profile_enable(False)

    verified = None
    if args.verify:
        flags["keep"] = True
        step()
        fence()
        if rank == 0:
            import numpy as np
            allreads = torch.empty(world * n * STRIDE, dtype=torch.uint8, device=dev)
            capi.synth_rows_device(local_rank, stream, SEED, 0, world * n, READ_LEN, STRIDE, allreads.data_ptr())
            torch.cuda.synchronize()
            eng.begin(capi.MK_MODE_KOC)
            eng.push_reads_device(allreads.data_ptr(), STRIDE, world * n, 0)
            single = eng.finish()
            verified = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(single, result["sketch"]))
            if not verified:
                a, b = single[0], result["sketch"][0]
                ka = np.sort(a[0].astype(np.uint64) << np.uint64(16) | a[1].astype(np.uint64))
                kb = np.sort(b[0].astype(np.uint64) << np.uint64(16) | b[1].astype(np.uint64))
                result["verify_detail"] = {"n_single": int(len(a[0])), "n_merged": int(len(b[0])),
                                           "same_multiset": bool(len(ka) == len(kb) and np.array_equal(ka, kb)),
                                           "first_diff": int(np.argmax(a[0][:min(len(a[0]), len(b[0]))] != b[0][:min(len(a[0]), len(b[0]))]))
                                           if len(a[0]) and len(b[0]) else -1,
                                           "only_single": int(len(np.setdiff1d(ka, kb))), "only_merged": int(len(np.setdiff1d(kb, ka)))}
            del allreads
        flags["keep"] = False

    if rank == 0:
        bases_per_step = float(world) * n * READ_LEN
        value = bases_per_step * args.steps / dt / 1e9
        # ---- roofline of the dominant kernel (mk_scan_kernel), per launch ----
        # algorithmic bytes (SURVEY.md 8d / DESIGN.md): 1 B per base scanned + 16 B per accepted k-mer occurrence
        # (slot read + slot write); the table clear/dump terms (16 S + 6 D) belong to the finish kernels.
        kmers = n * (READ_LEN - 21)
        accepted = kmers / 4096.0
        scan_bytes = n * READ_LEN + 16.0 * accepted
        scan_ms = prof["scan_ms"] / max(1, prof["scan_launches"])
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("reads_per_launch") == n:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Gbases/s sketched (150 bp synthetic reads, L3K11 -A)",
            "value": value, "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 k-mers over u8 bases (integer)", "data": "synthetic",
            "config": {"workload": "%d synthetic 150 bp reads per GPU resident in HBM (160 B rows), L3K11 .shuf "
                                   "{k=11,subk=6,drlevel=3}, -A counted sketch, begin+scan+finish per step" % n,
                       "reads_per_gpu": n, "read_len": READ_LEN, "row_stride": STRIDE,
                       "distinct_keys": result.get("distinct"), "parallelism": "reads sharded x%d, gather to rank 0" % world},
            "roofline": {"bound": "hbm", "kernel": "mk_scan_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": scan_bytes, "avg_launch_ms": scan_ms,
                         "launches": prof["scan_launches"]},
            "phases_ms_per_step": {"clear": prof["clear_ms"] / args.steps, "scan": prof["scan_ms"] / args.steps,
                                   "resolve": prof["resolve_ms"] / args.steps,
                                   "finish": prof["finish_ms"] / args.steps},
        }
        if verified is not None:
            line["merged_equals_single_engine"] = bool(verified)
            if "verify_detail" in result:
                line["verify_detail"] = result["verify_detail"]
        if args.backend != "nccl":
            line["config"]["parallelism"] += " (debug transport: %s%s)" % (args.backend, ", same device" if args.same_device else "")
        if world == 1 and not args.no_cpu_baseline:
            m = min(args.cpu_sample_reads, n)
            eng.begin(capi.MK_MODE_KOC)
            eng.push_reads_device(reads.data_ptr(), STRIDE, m, 0)
            sk = eng.finish()
            try:
                line["cpu_baseline"] = cpu_baseline(shuf, m, sk)
            except Exception as ex:  # a missing zcat etc. must not lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "Gbases/s", "cores": os.cpu_count(), "kind": "reference",
                                        "sample": "failed: %s" % ex}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
