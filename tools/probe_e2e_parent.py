#!/usr/bin/env python3
"""tools/probe_e2e_parent.py -- does the STATE OF THE PARENT PROCESS change what the FASTQ command line takes from process start?
bench.py's t_e2e leg runs the command line as a child of a process that has a HIP context, GPU memory and (cached by torch) pinned host
memory; tools/bench_e2e.py runs it from a bare interpreter.  Modes take turns on one box (repetitions outermost):
  bare     nothing but the interpreter
  ctx      torch.cuda initialised, a 1 GB device tensor alive
  pinned   ctx + 11 GB of pinned host memory allocated, dropped and left in torch's host cache
  released the same, cache emptied (torch._C._host_emptyCache)
Each mode is its own child python (the state cannot be undone), which runs the command line `--runs` times and prints `written` of each."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mode, fq, sp, runs):
    hold = []
    if mode != "bare":
        import torch
        torch.cuda.init()
        hold.append(torch.zeros(1 << 30, dtype=torch.uint8, device="cuda"))
        if mode in ("pinned", "released"):
            a = torch.empty(8 << 30, dtype=torch.uint8, pin_memory=True)
            b = torch.empty(3 << 30, dtype=torch.uint8, pin_memory=True)
            a.fill_(1); b.fill_(1)
            del a, b
            if mode == "released":
                torch._C._host_emptyCache()
        torch.cuda.synchronize()
    cli = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
    out = []
    for i in range(runs):
        time.sleep(2.0)
        d = tempfile.mkdtemp(prefix="o_", dir=os.path.dirname(fq))
        r = subprocess.run([cli, "dist", "-L", sp, "-A", "-o", d, "--quiet", "--timing", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        tm = {}
        for ln in r.stdout.decode(errors="replace").splitlines():
            if ln.startswith('{"timing"'):
                tm = json.loads(ln)["timing"]
        out.append({k: tm.get(k) for k in ("written", "hip_ready", "engine_ready", "last_push", "stream_wait_frame_s", "pin_s", "pinned_mib")})
        subprocess.run(["rm", "-rf", d])
    print(json.dumps({"mode": mode, "runs": out}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--fq"); ap.add_argument("--sp")
    ap.add_argument("--runs", type=int, default=4)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--modes", default="bare,ctx,pinned,released")
    a = ap.parse_args()
    if a.child:
        return child(a.child, a.fq, a.sp, a.runs)
    from metakssd_amd import capi
    tmp = tempfile.mkdtemp(prefix="mkpe_", dir="/dev/shm")
    try:
        fq, sp = os.path.join(tmp, "reads.fq"), os.path.join(tmp, "L3K11.shuf")
        capi.Shuf.generate(11, 6, 3, 11).write(sp)
        assert capi.lib.mk_synth_fastq_write_mt(fq.encode(), 20261002, 0, 50_000_000, 150, min(os.cpu_count() or 1, 64)) == 0
        cli = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
        subprocess.run([cli, "dist", "-L", sp, "-A", "-o", os.path.join(tmp, "warm"), "--quiet", fq], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for rep in range(a.reps):
            for mode in a.modes.split(","):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, "--fq", fq, "--sp", sp, "--runs", str(a.runs)],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                for ln in r.stdout.decode().splitlines():
                    if ln.startswith("{"):
                        j = json.loads(ln)
                        print("%-9s written %s | wait_frame %s | hip %s" % (j["mode"], " ".join("%.3f" % (x["written"] or 0) for x in j["runs"]),
                                                                         " ".join("%.3f" % (x["stream_wait_frame_s"] or 0) for x in j["runs"]),
                                                                         " ".join("%.3f" % (x["hip_ready"] or 0) for x in j["runs"])), flush=True)
                if r.returncode:
                    print(mode, "failed:", r.stderr.decode()[-300:])
    finally:
        subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
