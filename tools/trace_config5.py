#!/usr/bin/env python3
"""tools/trace_config5.py -- where a genome-directory run (BASELINE config 5) spends the window in which batches are on the device:
`rocprofv3 --kernel-trace -- metakssd dist -L <shuf> -o out <dir of genomes>` on the bench's 1 024 synthetic genomes, then per batch (one
mk_scan_packed_kernel each): when its scan ran, how long the other kernels of the batch took, and how long the queue was EMPTY between
kernels -- the link carries rows only while a scan kernel runs.

    python3 tools/trace_config5.py [--geometry L2K11] [--genomes 1024] [--flags "--batch-files 64"]"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--geometry", default="L2K11", choices=["L2K11", "L3K10"])
    ap.add_argument("--genomes", type=int, default=1024)
    ap.add_argument("--flags", default="", help="extra flags for the command line")
    ap.add_argument("--keep-csv", default="", help="copy the raw kernel trace here")
    a = ap.parse_args()
    import bench
    from metakssd_amd import capi
    tmp = tempfile.mkdtemp(prefix="mkc5t_", dir="/dev/shm")
    try:
        gd = os.path.join(tmp, "genomes")
        bases_each, _ = bench.write_genomes(gd, a.genomes, 4.0)
        k, sk, l, seed = {"L3K10": (10, 6, 3, 10), "L2K11": (11, 5, 2, 211)}[a.geometry]
        sp = os.path.join(tmp, a.geometry + ".shuf")
        capi.Shuf.generate(k, sk, l, seed).write(sp)
        cli = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
        subprocess.run([cli, "dist", "-L", sp, "-o", os.path.join(tmp, "warm"), "--quiet", gd] , check=True, stdout=subprocess.DEVNULL)  # page cache
        pd = os.path.join(tmp, "prof")
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", pd, "--", cli, "dist", "-L", sp, "-o", os.path.join(tmp, "out"),
                            "--quiet", "--slow-exit"] + a.flags.split() + [gd], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise SystemExit("the traced run failed: " + r.stderr.decode(errors="replace")[-400:])
        f = glob.glob(os.path.join(pd, "**", "*kernel_trace.csv"), recursive=True)[0]
        if a.keep_csv:
            shutil.copyfile(f, a.keep_csv)
        rows = [(x["Kernel_Name"].split("(")[0].replace("void ", ""), int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in csv.DictReader(open(f))]
        rows.sort(key=lambda x: x[1])
        scans = [i for i, x in enumerate(rows) if "mk_scan_packed_kernel" in x[0]]
        if not scans:
            raise SystemExit("no mk_scan_packed_kernel in the trace")
        t0, t1 = rows[scans[0]][1], max(x[2] for x in rows[scans[0]:])
        window = (t1 - t0) / 1e6
        scan_ms = sum(rows[i][2] - rows[i][1] for i in scans) / 1e6
        other, idle, per = {}, 0.0, []
        busy_until = rows[scans[0]][1]
        for x in rows[scans[0]:]:
            if x[1] > busy_until:
                idle += (x[1] - busy_until) / 1e6
            busy_until = max(busy_until, x[2])
            if "mk_scan_packed_kernel" not in x[0]:
                other[x[0]] = other.get(x[0], 0.0) + (x[2] - x[1]) / 1e6
        for j, i in enumerate(scans[:6] + scans[-2:]):
            nxt = rows[scans[scans.index(i) + 1]][1] if scans.index(i) + 1 < len(scans) else t1
            per.append({"scan_start_ms": round((rows[i][1] - t0) / 1e6, 3), "scan_ms": round((rows[i][2] - rows[i][1]) / 1e6, 3),
                        "until_next_scan_ms": round((nxt - rows[i][2]) / 1e6, 3)})
        row_bytes = a.genomes * 64.0 * (bases_each / float(241 - 2 * k) + 2.0)
        print(json.dumps({"geometry": a.geometry, "genomes": a.genomes, "flags": a.flags, "batches": len(scans), "window_ms": round(window, 3),
                          "scan_ms_total": round(scan_ms, 3), "other_kernels_ms_total": round(sum(other.values()), 3), "queue_empty_ms_total": round(idle, 3),
                          "other_kernels_ms": {k_: round(v, 3) for k_, v in sorted(other.items(), key=lambda kv: -kv[1])},
                          "link_gb_s_over_window": round(row_bytes / window / 1e6, 2), "link_gb_s_while_scanning": round(row_bytes / scan_ms / 1e6, 2),
                          "first_and_last_batches": per,
                          "what": "window = first scan kernel's start .. last kernel's end, under rocprofv3 --kernel-trace (the tracing itself stretches "
                                  "the gaps between kernels); rows cross the link only while a scan kernel runs"}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
