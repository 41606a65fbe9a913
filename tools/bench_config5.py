#!/usr/bin/env python3
"""BASELINE config 5: reference-genome directory sketch (no -A), L3K10 and L2K11, multi-FASTA input.
G synthetic genomes of MB megabases each (2 contigs, 70-column lines) in /dev/shm; times the product CLI and,
on a few genomes, the compiled reference (oracle/_ref/metakssd) when present."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from metakssd_amd import capi
G = int(os.environ.get("GENOMES", "64")); MB = float(os.environ.get("MBASES", "4"))
d = tempfile.mkdtemp(prefix="mkc5_", dir="/dev/shm")
gd = os.path.join(d, "genomes"); os.makedirs(gd)
rs = np.random.RandomState(5)
acgt = np.frombuffer(b"ACGT", np.uint8)
t0 = time.perf_counter()
for i in range(G):
    n = int(MB * 1e6)
    seq = acgt[rs.randint(0, 4, size=n)]
    with open(os.path.join(gd, "g%03d.fna" % i), "wb") as f:
        for c, (a, b) in enumerate(((0, n // 2), (n // 2, n))):
            f.write(b">g%d_contig%d\n" % (i, c))
            body = seq[a:b]
            pad = (-len(body)) % 70
            rows = np.concatenate([body, np.full(pad, ord("A"), np.uint8)]).reshape(-1, 70)
            out = np.concatenate([rows, np.full((rows.shape[0], 1), 10, np.uint8)], axis=1).reshape(-1)
            f.write(out[: len(out) - pad - (1 if pad else 0)].tobytes() + (b"\n" if pad else b""))
print("wrote %d genomes x %.1f Mbases in %.1f s" % (G, MB, time.perf_counter() - t0))
cli = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
ref = os.path.join(ROOT, "oracle", "_ref", "metakssd")
for name, (k, s, l, seed) in {"L3K10": (10, 6, 3, 10), "L2K11": (11, 5, 2, 211)}.items():
    sp = os.path.join(d, name + ".shuf"); capi.Shuf.generate(k, s, l, seed).write(sp)
    for rep in range(2):
        t0 = time.perf_counter()
        out = subprocess.check_output([cli, "dist", "-L", sp, "-p", os.environ.get("THREADS", "8"), "-o", os.path.join(d, "out_%s_%d" % (name, rep)), "--quiet", "--timing", gd])
        dt = time.perf_counter() - t0
        tl = [json.loads(ln)["timing"] for ln in out.decode(errors="replace").splitlines() if ln.startswith('{"timing"')]
        if tl:  # where the main thread's time goes: finish_s = sum of the mk_sketch_finish calls (each ends in a synchronisation)
            print("  timeline: engine ready %.3f, written %.3f, in finish calls %.3f s" % (tl[0]["engine_ready"], tl[0]["written"], tl[0]["finish_s"]))
        print("%s product CLI rep %d: %.2f s for %d genomes = %.1f genomes/s, %.2f Gbases/s" % (name, rep, dt, G, G / dt, G * MB / 1e3 / dt))
        print(json.dumps({"tool": "tools/bench_config5.py", "shuf": name, "who": "product CLI", "rep": rep, "genomes": G, "mbases_each": MB,
                          "threads": int(os.environ.get("THREADS", "8")), "seconds": round(dt, 3), "genomes_per_s": round(G / dt, 1),
                          "gbases_per_s": round(G * MB / 1e3 / dt, 3)}))
    if os.path.exists(ref):
        few = sorted(os.listdir(gd))[:int(os.environ.get("REF_GENOMES", "64"))]
        sub = os.path.join(d, "few_" + name); os.makedirs(sub)
        for f in few: os.symlink(os.path.join(gd, f), os.path.join(sub, f))
        t0 = time.perf_counter()
        subprocess.run([ref, "dist", "-L", sp, "-p", str(os.cpu_count()), "-o", os.path.join(d, "ref_" + name), sub], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        print("%s compiled reference (-p %d): %.2f s for %d genomes = %.2f genomes/s" % (name, os.cpu_count(), dt, len(few), len(few) / dt))
        print(json.dumps({"tool": "tools/bench_config5.py", "shuf": name, "who": "compiled reference -p %d" % os.cpu_count(), "genomes": len(few),
                          "mbases_each": MB, "seconds": round(dt, 3), "genomes_per_s": round(len(few) / dt, 2)}))
subprocess.call(["rm", "-rf", d])
