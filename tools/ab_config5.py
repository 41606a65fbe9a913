#!/usr/bin/env python3
"""tools/ab_config5.py -- BASELINE config 5 through the command line with variants taking turns on ONE box: every variant is a set of
environment variables and extra flags; bench.py's own leg (1 024 genomes, 5 runs after a warm-up, the parent's clock) is run for each,
twice round, and the walls + the PCIe roofline entry of each are printed as JSON lines.

    python3 tools/ab_config5.py --geometry L2K11 --variant "default" --variant "MK_BATCH_TAB_BITS=16" --variant "flags:--batch-mib 256" """
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--geometry", default="L2K11")
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    import bench
    from metakssd_amd import capi
    variants = a.variant or ["default"]
    for rnd in range(a.rounds):
        for v in variants:
            env_add, flags = {}, []
            for part in v.split(";"):
                part = part.strip()
                if part.startswith("flags:"):
                    flags += part[6:].split()
                elif "=" in part:
                    k, val = part.split("=", 1)
                    env_add[k] = val
            old = {k: os.environ.get(k) for k in env_add}
            os.environ.update(env_add)
            try:
                r = bench.leg_config5(capi, ref_genomes=0, extra_flags=flags, only=a.geometry)
            finally:
                for k, val in old.items():
                    if val is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = val
            g = r.get(a.geometry, {})
            g.pop("set_union", None)
            print(json.dumps({"variant": v, "round": rnd, "geometry": a.geometry, "seconds": g.get("seconds"), "all_runs_s": g.get("all_runs_s"),
                              "genomes_per_s": g.get("genomes_per_s"), "engine_ready_s": g.get("engine_ready_s"), "written_s": g.get("written_s"),
                              "window_s": (g.get("roofline") or {}).get("window_s"), "pcie_frac": (g.get("roofline") or {}).get("frac"),
                              "batches": g.get("batches")}), flush=True)


if __name__ == "__main__":
    main()
