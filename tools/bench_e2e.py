#!/usr/bin/env python3
"""tools/bench_e2e.py -- t_e2e sweeps: `metakssd dist -L L3K11.shuf -A` on a synthetic FASTQ in /dev/shm for several front-end
thread counts (-p) and chunk sizes; one JSON line per run and a summary line (kept under profiles/ by the caller).
    python tools/bench_e2e.py [--reads 50000000] [--threads 8,16,32,64] [--chunks 8] [--reps 2] [--out gpurun_out/e2e.json]"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--threads", default="8,16,32,48,64")
    ap.add_argument("--chunks", default="8")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default=None)
    ap.add_argument("--variants", default="", help="';'-separated extra CLI flag sets to run each configuration with, e.g. ';--keep-pages'")
    a = ap.parse_args()
    from metakssd_amd import capi
    cores = os.cpu_count() or 1
    tmp = tempfile.mkdtemp(prefix="mke2e_", dir="/dev/shm")
    runs = []
    try:
        fq, sp = os.path.join(tmp, "reads.fq"), os.path.join(tmp, "L3K11.shuf")
        capi.Shuf.generate(11, 6, 3, 11).write(sp)
        t0 = time.perf_counter()
        assert capi.lib.mk_synth_fastq_write_mt(fq.encode(), 20261002, 0, a.reads, 150, min(cores, 64)) == 0
        print("wrote %.2f GB in %.1f s (%d cores)" % (os.path.getsize(fq) / 1e9, time.perf_counter() - t0, cores), flush=True)
        cli = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
        ref_hash = None
        combos = [(chunk, T, v) for v in a.variants.split(";") for chunk in [int(x) for x in a.chunks.split(",")]
                  for T in [int(x) for x in a.threads.split(",")]]
        for rep in range(a.reps):  # repetitions outermost: the configurations take turns, so that a drifting box touches all alike
            for chunk, T, variant in combos:
                if True:
                    time.sleep(1.5)  # (the driver is still taking the previous process's device memory back)
                    out = os.path.join(tmp, "out")
                    shutil.rmtree(out, ignore_errors=True)
                    m0 = time.monotonic()
                    t0 = time.perf_counter()
                    r = subprocess.run([cli, "dist", "-L", sp, "-A", "-o", out, "--quiet", "--timing", "-p", str(T), "--chunk-mib",
                                        str(chunk)] + variant.split() + [fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                    wall = time.perf_counter() - t0
                    m1 = time.monotonic()
                    tm = {}
                    for ln in r.stdout.decode(errors="replace").splitlines():
                        if ln.startswith('{"timing"'):
                            tm = json.loads(ln)["timing"]
                    h = subprocess.run("cat %s/combco.0 %s/combco.0.a | sha256sum" % (out, out), shell=True, stdout=subprocess.PIPE).stdout.decode().split()[0]
                    ref_hash = ref_hash or h
                    rec = {"p": T, "chunk_mib": chunk, "flags": variant, "rep": rep, "rc": r.returncode, "wall_s": round(wall, 4),
                           "gbases_s_wall": round(a.reads * 150 / wall / 1e9, 2),
                           "gbases_s_minus_init": round(a.reads * 150 / max(wall - tm.get("hip_ready", 0), 1e-9) / 1e9, 2),
                           "spawn_s": round(tm.get("t0_abs", m0) - m0, 4), "exit_s": round(m1 - tm.get("exit_abs", m1), 4),
                           "same_sketch": h == ref_hash, "timing": tm}
                    if r.returncode:
                        rec["stderr"] = r.stderr.decode(errors="replace")[-300:]
                    runs.append(rec)
                    print(json.dumps(rec), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    best = min((r for r in runs if r["rc"] == 0), key=lambda r: r["wall_s"], default=None)
    summary = {"tool": "tools/bench_e2e.py", "reads": a.reads, "file_gb": round(a.reads * 318.9 / 1e9, 2), "cores": cores, "best": best,
               "runs": runs}
    print("SUMMARY " + json.dumps({k: summary[k] for k in ("reads", "cores", "best")}), flush=True)
    if a.out:
        json.dump(summary, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
