#!/usr/bin/env python3
"""tools/probe_numa.py -- where the FASTQ file's pages, the framers and the GPU sit (round 5: why the FASTQ front end stops at ~150 GB/s
of text whatever the thread count).  Prints the host's NUMA layout, the node of the GPU, the nodes the synthetic FASTQ's tmpfs pages
landed on (numa_maps of a populated mapping), and the command line's timeline with its threads confined to one node at a time
(taskset), to both, and left alone."""
import glob
import json
import mmap
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sh(cmd):
    return subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors="replace").strip()


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        n = int(d.rsplit("node", 1)[1])
        nodes[n] = {"cpus": open(d + "/cpulist").read().strip(),
                    "mem": " ".join(l.split(":")[1].strip() for l in open(d + "/meminfo") if "MemTotal" in l or "MemFree" in l)}
    print(json.dumps({"nodes": nodes}))
    for d in glob.glob("/sys/class/drm/card*/device"):
        try:
            print(json.dumps({"gpu": d, "numa_node": open(d + "/numa_node").read().strip(), "vendor": open(d + "/vendor").read().strip()}))
        except OSError:
            pass
    print(json.dumps({"thp": sh("cat /sys/kernel/mm/transparent_hugepage/enabled"), "shmem_thp": sh("cat /sys/kernel/mm/transparent_hugepage/shmem_enabled"),
                      "shm": sh("df -h /dev/shm | tail -1"), "taskset": sh("which taskset"), "kernel": sh("uname -r")}))
    from metakssd_amd import capi
    d = "/dev/shm/mk_numa"
    os.makedirs(d, exist_ok=True)
    fq, sp = d + "/reads.fq", d + "/L3K11.shuf"
    capi.Shuf.generate(11, 6, 3, 11).write(sp)
    assert capi.lib.mk_synth_fastq_write_mt(fq.encode(), 20261002, 0, reads, 150, min(os.cpu_count() or 1, 64)) == 0
    with open(fq, "rb") as f:
        m = mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ, flags=mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0))
        for ln in open("/proc/self/numa_maps"):
            if "reads.fq" in ln:
                print(json.dumps({"file_pages_by_node": dict(re.findall(r"N(\d+)=(\d+)", ln)), "line": ln.strip()[:300]}))
        m.close()
    cli = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
    variants = [("free", [])]
    if sh("which taskset"):
        for n, v in nodes.items():
            variants.append(("node%d" % n, ["taskset", "-c", v["cpus"]]))
    for rep in range(3):
        for name, pre in variants:
            for flags in ([], ["--mmap-input"]):
                time.sleep(1.5)
                out = d + "/out"
                subprocess.run(["rm", "-rf", out])
                t0 = time.monotonic()
                r = subprocess.run(pre + [cli, "dist", "-L", sp, "-A", "-o", out, "--quiet", "--timing"] + flags + [fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                wall = time.monotonic() - t0
                tm = {}
                for ln in r.stdout.decode(errors="replace").splitlines():
                    if ln.startswith('{"timing"'):
                        tm = json.loads(ln)["timing"]
                print(json.dumps({"run": name, "flags": flags, "rep": rep, "rc": r.returncode, "wall": round(wall, 4),
                                  **{k: tm.get(k) for k in ("hip_ready", "engine_ready", "first_push", "last_push", "written", "stream_wait_frame_s", "wait_call_s", "threads")}}), flush=True)
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
