#!/usr/bin/env python3
"""BASELINE config 5 through the command line with other flags than bench.py's leg uses (same genomes, same timing rules):
   python tools/bench_config5_variants.py            # L3K10 and L2K11 with one to four engines in turn (MK_C5_ENGINES=1,2 to choose)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from metakssd_amd import capi
ENGINES = [int(x) for x in os.environ.get("MK_C5_ENGINES", "1,2,3,4").split(",")]
for name, flags in [(g, ["--engines", str(n)]) for g in ("L3K10", "L2K11") for n in ENGINES]:
    r = bench.leg_config5(capi, ref_genomes=0, extra_flags=flags, only=name)
    print(json.dumps({"tool": "tools/bench_config5_variants.py", "shuf": name, "flags": flags, **{k: r[name].get(k) for k in
          ("genomes_per_s", "genomes_per_s_after_start", "seconds", "all_runs_s", "finish_ms_per_genome", "engine_ready_s")}}))
