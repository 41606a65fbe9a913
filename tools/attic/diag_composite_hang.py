#!/usr/bin/env python3
"""diagnostic (GPU box): the first step of the composite_mix_L2K11 golden case (three FASTA references, L2K11, -p 4) under a
timeout; when it hangs, the same command under rocgdb with a SIGINT after 60 s and the backtraces of all threads"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc
d = tempfile.mkdtemp(prefix="diag_")
refs, qry = gc.build_composite_inputs("composite_mix_L2K11", d)
sp = os.path.join(d, "L2K11.shuf"); gc.make_shuf("L2K11", sp)
cli = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
for extra in ([], ["--host-fasta"]):
    cmd = [cli, "dist", "-L", sp, "-p", "4"] + extra + ["-o", os.path.join(d, "sk" + ("h" if extra else "")), ] + refs
    t0 = time.time()
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=90)
        print("dist %s: rc %d in %.1f s" % (extra, r.returncode, time.time() - t0), r.stderr.decode()[-300:])
    except subprocess.TimeoutExpired:
        print("dist %s: HANG (90 s). backtraces:" % extra)
        g = subprocess.run(["timeout", "-s", "INT", "60", "rocgdb", "-batch", "-ex", "run", "-ex", "thread apply all bt 12", "--args"] + cmd,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        print(g.stdout.decode(errors="replace")[-6000:])
