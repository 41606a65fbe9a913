#!/usr/bin/env python3
"""BASELINE config 5 through the command line with other flags than bench.py's leg uses (same genomes, same timing rules):
   python tools/bench_config5_variants.py            # L2K11 with one and with two engines, L3K10 with one and two"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from metakssd_amd import capi
for name, flags in (("L2K11", ["--engines", "1"]), ("L2K11", ["--engines", "2"]), ("L3K10", ["--engines", "1"]), ("L3K10", ["--engines", "2"])):
    r = bench.leg_config5(capi, ref_genomes=0, extra_flags=flags, only=name)
    print(json.dumps({"tool": "tools/bench_config5_variants.py", "shuf": name, "flags": flags, **{k: r[name].get(k) for k in
          ("genomes_per_s", "genomes_per_s_after_start", "seconds", "all_runs_s", "finish_ms_per_genome", "engine_ready_s")}}))
